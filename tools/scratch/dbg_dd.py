import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from chemlab_amd import workloads as W
from chemlab_amd.engine import Engine
from oracle.oracle import OracleEngine
for skin in (0.0, 0.55):
    spec = W.reactive_melt(n=8788, seed=91, interval=10)
    spec["rebuild_criterion"] = 0
    for r in spec["reaction"]["reactions"]: r["rate"] = 1e9
    g = Engine(device=0, precision=64); o = OracleEngine(); s = Engine(device=0, precision=64)
    g.set_option("dd_self", 1); g.set_option("list_skin", skin); s.set_option("list_skin", skin)
    W.apply(spec, g); W.apply(spec, o); W.apply(spec, s)
    for k in range(6):
        g.run(5); o.run(5); s.run(5)
        print(skin, k, "dd", g.timers()["rebuilds"], g.timers()["list_rebuilds"], "oracle", o.timers()["rebuilds"], "single", s.timers()["rebuilds"], s.timers()["list_rebuilds"], flush=True)
    g.close(); o.close(); s.close()
