#!/usr/bin/env python3
"""Diagnostic: per-tile phase stamps of k_pair_tiles (100 MHz wall clock)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, '.')
from chemlab_amd import workloads as W
from chemlab_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spec = W.reactive_melt(n=n, rho=0.8, seed=2)
e = Engine(precision=32)
W.apply(spec, e)
for kv in sys.argv[2:]:
    k, v = kv.split('='); e.set_option(k, float(v))
e.reactions_enable(False); e.run(int(__import__("os").environ.get("EQUIL", "50")))
e.set_option("debug_stamps", 1)
e.run(3)
e.sync()
lib = e.api.lib
lib.chem_debug_dump.restype = C.c_int64
buf = np.zeros(6 * 20000, dtype=np.int64)
m = lib.chem_debug_dump(C.c_void_p(e.ctx), buf.ctypes.data_as(C.c_void_p), buf.size)
d = buf[:m].reshape(-1, 6)
d = d[d[:, 0] > 0]
t0 = d[:, 0].min()
us = lambda x: x / 100.0
print("blocks", len(d), "kernel span us", us(d[:, 3].max() - t0))
print("desc  us: mean %.2f p50 %.2f p90 %.2f" % (us((d[:,1]-d[:,0]).mean()), us(np.median(d[:,1]-d[:,0])), us(np.percentile(d[:,1]-d[:,0], 90))))
print("fill  us: mean %.2f p50 %.2f p90 %.2f" % (us((d[:,2]-d[:,1]).mean()), us(np.median(d[:,2]-d[:,1])), us(np.percentile(d[:,2]-d[:,1], 90))))
print("loop  us: mean %.2f p50 %.2f p90 %.2f" % (us((d[:,3]-d[:,2]).mean()), us(np.median(d[:,3]-d[:,2])), us(np.percentile(d[:,3]-d[:,2], 90))))
print("block us: mean %.2f" % us((d[:,3]-d[:,0]).mean()))
st = np.sort(d[:, 0] - t0)
print("block start times us (deciles):", [round(us(x), 1) for x in np.percentile(st, [0, 10, 25, 50, 75, 90, 100])])
conc = [(np.sum((d[:,0] <= t) & (d[:,3] > t))) for t in np.linspace(d[:,0].min(), d[:,3].max(), 12)]
print("concurrent blocks over time:", conc)
