#!/usr/bin/env python3
"""Steps/s of the BASELINE.json parity configurations C2-C4 on the GPU next to the single-thread oracle
(documentation only; the bench line is bench.py's C5).   python tools/bench_configs.py"""
import sys, time
sys.path.insert(0, '.')
from chemlab_amd import workloads as W
from chemlab_amd.engine import Engine
from oracle.oracle import OracleEngine

CFG = [("C2 32k LJ melt, no reactions", W.lj_melt(n=32000, rho=0.8, seed=21), dict(reactions=False), 4000, 40),
       ("C3 128k tabulated polymer melt (bonds, angles)", W.polymer_melt(n_chains=4000, chain_len=32, seed=3), dict(), 2000, 10),
       ("C4 256k reactive LJ (reactions every 500 steps)", W.reactive_melt(n=256000, rho=0.8, seed=5, interval=500), dict(), 3000, 10)]
for name, spec, kw, nsteps, csteps in CFG:
    g = Engine(precision=32); W.apply(spec, g, **kw)
    g.run(200); g.sync()
    t0 = time.perf_counter(); g.run(nsteps); g.sync(); tg = time.perf_counter() - t0
    g.close()
    o = OracleEngine(); W.apply(spec, o, **kw)
    o.run(2)
    t0 = time.perf_counter(); o.run(csteps); to = time.perf_counter() - t0
    print("%-52s GPU %9.0f steps/s   oracle (1 thread) %7.2f steps/s   x%.0f" % (name, nsteps / tg, csteps / to, (nsteps / tg) / (csteps / to)))
