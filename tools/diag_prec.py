import sys, numpy as np
sys.path.insert(0, '.')
from chemlab_amd import workloads as W
from chemlab_amd.engine import Engine
from oracle.oracle import OracleEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
spec = W.lj_melt(n=n, rho=0.8, seed=21, gamma=1.0)
m = Engine(precision=32); W.apply(spec, m); m.run(400)
spec = dict(spec, pos=m.get_state("POS"), vel=m.get_state("VEL"))
m.close()
o = OracleEngine(); W.apply(spec, o, thermostat=False); o.run(0); fo = o.get_state("FORCE")
L = spec["box"][0]
for name, prec, opts in (("g32 tiles", 32, {}), ("g32 per-cell", 32, {"tiles": 0}), ("g64 tiles", 64, {})):
    g = Engine(precision=prec); W.apply(spec, g, thermostat=False)
    for k, v in opts.items(): g.set_option(k, v)
    g.run(0)
    f = g.get_state("FORCE"); p = g.get_state("POS")
    err = np.abs(f - fo).max(1)
    w = int(err.argmax())
    print("%-14s max|dF|/max|F| %.3e  mean|dF| %.3e  max|F| %.1f  pos roundtrip %.3e" % (name, err.max() / np.abs(fo).max(), err.mean(), np.abs(fo).max(), np.abs(p - spec["pos"]).max()))
    d = spec["pos"] - spec["pos"][w]; d -= L * np.rint(d / L); r = np.sqrt((d * d).sum(1)); r[w] = 9
    print("   worst particle", w, "pos", spec["pos"][w], "cell frac", (spec["pos"][w] / (L / int(L // 2.8))) % 1.0, "nearest r %.4f" % r.min(), "F", fo[w], "dF", (f - fo)[w])
    # error vs position in box: worst 200 particles' coordinates quantiles
    idx = np.argsort(err)[-200:]
    print("   worst-200 mean |x|:", np.abs(spec["pos"][idx]).mean(0), " all:", np.abs(spec["pos"]).mean(0), " err of worst-200 min %.2e" % err[idx].min())
    g.close()
