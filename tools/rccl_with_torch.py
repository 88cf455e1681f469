#!/usr/bin/env python3
"""The production communicator call (chem_comm_init -> RCCL) inside a process that has torch, gloo and a CUDA(HIP) tensor alive --
what every rank of `bench.py --gpus N` / the multi-process driver looks like.  One rank (RCCL refuses two on a device): checks that
the library's dlopen of librccl and torch's own copy coexist (they resolve to ONE shared object, printed at the end) and that the
slab machinery runs over it.   usage (on an MI355X): python tools/rccl_with_torch.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group(backend="gloo")
torch.cuda.synchronize(0)
x = torch.ones(4, device="cuda:0"); print("torch on gpu ok", float(x.sum()))
from chemlab_amd import workloads as W
from chemlab_amd.engine import Engine, comm_unique_id
e = Engine(device=0, precision=32)
uid = comm_unique_id()
e.comm_init(1, 0, uid)          # the production call: RCCL communicator (one rank), slab machinery on
spec = W.reactive_melt(n=32768, seed=3, interval=20)
W.apply(spec, e)
e.run(60); e.sync()
print("rccl world-of-one run ok: events", len(e.get_events()), "rebuilds", e.timers()["rebuilds"])
import ctypes
print([l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l and "r-xp" in l])
