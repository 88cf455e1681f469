"""Child process of test_dd_eight_slabs_at_bench_size_in_process (tests/test_gpu_parity.py): the N = 8 slab
decomposition of the bench workload with eight in-process ranks on one GPU against a single-domain engine."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from chemlab_amd import workloads as W          # noqa: E402
from chemlab_amd.engine import Engine           # noqa: E402
from helpers import sorted_events               # noqa: E402


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def main():
    P = 8
    spec = W.reactive_melt(n=1000000, rho=0.8, seed=2, interval=20)
    s = Engine(device=0, precision=32)
    W.apply(spec, s, thermostat=False)
    s.run(45)
    ref = sorted_events(s.get_events())
    xs = s.get_state("POS_UNFOLDED")
    engs = [Engine(device=0, precision=32) for _ in range(P)]
    out, err = [None] * P, [None] * P

    def rank(r):
        try:
            g = engs[r]
            g.comm_init_local(P, r, 777)
            W.apply(spec, g, thermostat=False)
            g.run(45)
            x = g.get_state("POS_UNFOLDED")                      # a collective on the decomposed path: every rank calls it
            out[r] = dict(ev=sorted_events(g.get_events()), x=x if r == 0 else None, reb=g.timers()["rebuilds"])
        except BaseException as e:   # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=rank, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    for e in err:
        if e is not None:
            raise e
    assert len(ref) > 200000
    for r in range(P):
        assert [e[:4] for e in out[r]["ev"]] == [e[:4] for e in ref]
        assert out[r]["reb"] >= 4
    assert rel_err(out[0]["x"], xs) < 1e-4
    for e in engs + [s]:
        e.close()
    print("EIGHT_SLABS_OK")


if __name__ == "__main__":
    main()
