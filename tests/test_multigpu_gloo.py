"""world_size-2 gloo test (CPU): the host-side plumbing of the N>1 path -- rendezvous, unique-id
broadcast, slab ownership, max-over-ranks timing.  The device data path (RCCL halo) is covered on
the GPU by the self / in-process multi-rank parity tests."""
import os
import subprocess
import sys

import numpy as np

from chemlab_amd import multigpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
from chemlab_amd import multigpu, workloads as W
dist = multigpu.init_process_group()
rank, world = dist.get_rank(), dist.get_world_size()
uid = multigpu.broadcast_bytes(dist, bytes(range(128)) if rank == 0 else None)
assert uid == bytes(range(128))
spec = W.reactive_melt(n=8788, seed=5)
own = multigpu.owner_of(spec["pos"][:, 2], spec["box"][2], spec["rc"], spec["skin"], world)
mine = int((own == rank).sum())
import torch
t = torch.tensor([mine], dtype=torch.int64)
dist.all_reduce(t)
assert int(t[0]) == spec["n"], (int(t[0]), spec["n"])          # every particle has exactly one owner
mx = multigpu.max_over_ranks(dist, 1.0 + rank)
assert mx == float(world)
dist.barrier()
# one file per rank: the two workers share a stdout pipe and their lines can interleave
with open(os.path.join(%r, 'rank_%%d.json' %% rank), 'w') as fh:
    json.dump(dict(rank=rank, mine=mine), fh)
'''


def test_slab_layers_cover_the_box():
    for nz, p in [(38, 8), (38, 4), (38, 2), (7, 3), (9, 4), (16, 8)]:
        lay = multigpu.slab_layers(nz, p)
        assert lay[0][0] == 0 and lay[-1][1] == nz
        assert all(a[1] == b[0] for a, b in zip(lay, lay[1:]))
        assert all(2 <= z1 - z0 for z0, z1 in lay)
        assert max(z1 - z0 for z0, z1 in lay) - min(z1 - z0 for z0, z1 in lay) <= 1
    assert multigpu.node_grid(8) == (1, 1, 8)


def test_owner_of_matches_layers():
    own = multigpu.owner_of(np.array([0.0, 5.59, 5.61, 21.0, -0.1, 22.0]), 21.84, 2.5, 0.3, 3)   # 7 layers of 3.12: (0,3),(3,5),(5,7)
    assert own.tolist() == [0, 0, 0, 2, 2, 0]


def test_two_rank_gloo_rendezvous_and_partition(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, str(tmp_path)))
    import socket
    with socket.socket() as sk:          # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    import json
    out = [json.load(open(tmp_path / ("rank_%d.json" % r))) for r in range(2)]
    assert [o["rank"] for o in out] == [0, 1]


def test_bench_started_plainly_spawns_its_ranks(tmp_path):
    """`python bench.py --gpus 2` (the driver's command line, no torch.distributed.run around it): the parent starts the
    ranks as a child process and relays rank 0's line and the exit code.  --rendezvous-only stops before any GPU work."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d == dict(rendezvous="ok", n_gpus=2, max_rank=1, node_grid=list(multigpu.node_grid(2)))
    # a failing rank's exit code comes back through the parent (no GPU here: chem_create fails loudly)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "8788", "--equil", "0", "--steps", "1",
                          "--warmup", "0", "--cpu-steps", "0", "--f64-steps", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    import torch
    if not torch.cuda.is_available():
        assert bad.returncode != 0


def test_only_the_root_rank_writes(tmp_path):
    """Multi-process driver runs (chemlab_amd/rank.py): every writer opens its files through wopen -- the real file on rank 0,
    the null device elsewhere; reads always pass through.  Started plainly, join_ranks() reports a world of one."""
    import os
    from chemlab_amd import rank, start_simulation
    p = str(tmp_path / "out.txt")
    try:
        rank.set_root(False)
        with rank.wopen(p, "w") as f:
            f.write("x")
        assert not os.path.exists(p) and not rank.is_root()
        rank.set_root(True)
        with rank.wopen(p, "a") as f:
            f.write("y")
        rank.set_root(False)
        with rank.wopen(p) as f:                       # reading is never redirected
            assert f.read() == "y"
    finally:
        rank.set_root(True)
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK") if k in os.environ}
    try:
        assert start_simulation.join_ranks() == (0, 1)
    finally:
        os.environ.update(env)
