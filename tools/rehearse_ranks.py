#!/usr/bin/env python3
"""Rehearsal of the P-rank slab decomposition on ONE GPU: all ranks as host threads of this process
(chem_comm_init_local, device-to-device exchange), the bench workload at full size.  Checks that the geometry, the
capacities and the collective flow hold at P ranks (P = 8: 4-5 cell layers per slab at 1M particles); the time is the
sum of all ranks' device work on one card, not a scaling measurement.
usage: rehearse_ranks.py [P=8] [n=1000000] [steps=200]"""
import sys, threading, time
import numpy as np
sys.path.insert(0, '.')
from chemlab_amd import workloads as W
from chemlab_amd.engine import Engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
spec = W.reactive_melt(n=n, rho=0.8, seed=2, interval=100)
engs = [Engine(precision=32) for _ in range(P)]
out, err = [None] * P, [None] * P
bar = threading.Barrier(P)

def rank(r):
    try:
        g = engs[r]
        g.comm_init_local(P, r, 4242)
        W.apply(spec, g)
        g.reactions_enable(False); g.run(100); g.reactions_enable(True)
        g.sync(); bar.wait()
        t0 = time.time()
        g.run(steps); g.sync()
        bar.wait()
        dt = time.time() - t0
        tm = g.timers()
        out[r] = dict(wall=dt, rebuilds=tm["rebuilds"], events=len(g.get_events()), ekin=g.observe()["ekin"])
    except BaseException as e:   # noqa: BLE001
        err[r] = e
        try: bar.abort()
        except Exception: pass

th = [threading.Thread(target=rank, args=(r,)) for r in range(P)]
for t in th: t.start()
for t in th: t.join(timeout=900)
for e in err:
    if e is not None: raise e
w = max(o["wall"] for o in out)
print("P=%d n=%d: %d steps in %.3f s (%.0f steps/s with all ranks on one card), rebuilds %s, events %s, ekin spread %.3g"
      % (P, n, steps, w, steps / w, [o["rebuilds"] for o in out][:3], out[0]["events"], np.ptp([o["ekin"] for o in out])))
