#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase stamps of the last rebuilding k_rebuild_fused launch (100 MHz wall clock)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, '.')
from chemlab_amd import workloads as W
from chemlab_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spec = W.reactive_melt(n=n, rho=0.8, seed=2)
e = Engine(precision=int(os.environ.get("PREC", "32")))
W.apply(spec, e)
for kv in sys.argv[2:]:
    k, v = kv.split('='); e.set_option(k, float(v))
e.reactions_enable(False); e.run(int(os.environ.get("EQUIL", "50")))
rs = int(os.environ.get("REACT_STEPS", "0"))
if rs:
    e.reactions_enable(True); e.run(rs)      # bonds, exclusions and bonded work lists in the rebuild
e.set_option("debug_stamps", 1)
e.run(int(os.environ.get("STAMP_STEPS", "30")))      # (long enough to hold a list build: they are ~13 steps apart with the list skin)
e.sync()
lib = e.api.lib
lib.chem_debug_dump_rebuild.restype = C.c_int64
buf = np.zeros(8 * 4096, dtype=np.int64)
m = lib.chem_debug_dump_rebuild(C.c_void_p(e.ctx), buf.ctypes.data_as(C.c_void_p), buf.size)
d = buf[:m].reshape(-1, 8)
d = d[d[:, 0] > 0]
t0 = d[:, 0].min()
us = lambda x: np.asarray(x) / 100.0
names = ["bin work", "barrier 1 wait", "sort work", "barrier 2 wait", "prefix+bonded prep", "tiles", ]
print("workgroups", len(d), " launch start spread us %.1f" % us(d[:, 0].max() - t0))
for k, nm in enumerate(names):
    dt = us(d[:, k + 1] - d[:, k])
    print("%-20s mean %7.1f  min %7.1f  p50 %7.1f  max %7.1f" % (nm, dt.mean(), dt.min(), np.median(dt), dt.max()))
tot = us(d[:, 7] >> 16); tiles = d[:, 7] & 0xffff
cb = tot - us(d[:, 6] - d[:, 0])
print("%-20s mean %7.1f  min %7.1f  p50 %7.1f  max %7.1f" % ("copy-back", cb.mean(), cb.min(), np.median(cb), cb.max()))
print("%-20s mean %7.1f  min %7.1f  p50 %7.1f  max %7.1f" % ("whole workgroup", tot.mean(), tot.min(), np.median(tot), tot.max()))
print("phase ends (us after first start): bin %.1f | bar1 %.1f | sort %.1f | bar2 %.1f | tiles %.1f | end %.1f" % tuple(
    us(x) for x in (d[:, 1].max() - t0, d[:, 2].max() - t0, d[:, 3].max() - t0, d[:, 4].max() - t0, d[:, 6].max() - t0, (d[:, 0] + (d[:, 7] >> 16)).max() - t0)))
print("tiles per workgroup: min %d p50 %d max %d; us per tile (mean over wgs) %.2f" % (tiles.min(), np.median(tiles), tiles.max(), (us(d[:, 6] - d[:, 5]) / np.maximum(tiles, 1)).mean()))
out = os.environ.get("STAMPS_NPY")
if out: np.save(out, d)
