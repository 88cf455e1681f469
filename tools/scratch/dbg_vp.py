import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from chemlab_amd import workloads as W
from chemlab_amd.engine import Engine
from oracle.oracle import OracleEngine
def rel(a, b): return np.abs(a - b).max() / np.abs(b).max()
for prec in (64, 32):
  for split in (0, 11):
    for inl in (1, 0):
        spec = W.reactive_melt(n=8788, seed=61, interval=20)
        for r in spec["reaction"]["reactions"]: r["rate"] = 1e9
        g = Engine(device=0, precision=prec); o = OracleEngine()
        W.apply(spec, g); W.apply(spec, o)
        g.set_option("tile_split", split); g.set_option("bonds_inline", inl)
        g.run(0); o.run(0)
        for _ in range(3): g.run(20); o.run(20)
        e0 = rel(g.get_state("FORCE"), o.get_state("FORCE"))
        vp = g.get_verlet_pairs()
        e1 = rel(g.get_state("FORCE"), o.get_state("FORCE"))
        g.run(0)
        e2 = rel(g.get_state("FORCE"), o.get_state("FORCE"))
        print(prec, split, inl, "before %.2e after_vp %.2e after_run0 %.2e" % (e0, e1, e2), flush=True)
        g.close(); o.close()
