import sys, numpy as np
sys.path.insert(0, '.')
from chemlab_amd import workloads as W
from chemlab_amd.engine import Engine
spec = W.reactive_melt(n=8788, seed=21, interval=20)
a, b = Engine(precision=32), Engine(precision=32)
for e in (a, b): W.apply(spec, e)
b.set_option("fused_rebuild", 0)
import os
if os.environ.get("TPP"):
    a.set_option("tpp", int(os.environ["TPP"])); b.set_option("tpp", int(os.environ["TPP"]))
for it in range(80):
    a.run(1); b.run(1)
    fa, fb = a.get_state("FORCE"), b.get_state("FORCE")
    pa, pb = a.get_state("POS"), b.get_state("POS")
    if not np.array_equal(fa, fb) or not np.array_equal(pa, pb):
        bad = np.where(np.any(fa != fb, axis=1))[0]
        print("step", it + 1, "force differs on", len(bad), "particles", bad[:10], "pos equal:", np.array_equal(pa, pb))
        print(fa[bad[:3]], fb[bad[:3]])
        print("rebuilds", a.timers()["rebuilds"], b.timers()["rebuilds"], "events", len(a.get_events()), len(b.get_events()))
        import ctypes as C
        def flist(e, tg):
            lib = e.api.lib; lib.chem_debug_force_list.restype = C.c_int64
            buf = np.zeros(512, dtype=np.int32)
            m = lib.chem_debug_force_list(C.c_void_p(e.ctx), C.c_int32(int(tg)), buf.ctypes.data_as(C.c_void_p), C.c_int64(512))
            return buf[:max(m, 0)]
        la, lb = flist(a, bad[0]), flist(b, bad[0])
        print("list lengths", len(la), len(lb), "same order:", np.array_equal(la, lb), "same set:", set(la) == set(lb))
        L = np.array(spec["box"])
        pos = pa
        def dist(i, j):
            d = pos[i] - pos[j]
            if L is not None: d -= L * np.round(d / L)
            return float(np.sqrt((d * d).sum()))
        for t in sorted(set(la) ^ set(lb)): print("  only in", "fused" if t in set(la) else "chain", t, "r now", dist(bad[0], t))
        k = next((k for k in range(min(len(la), len(lb))) if la[k] != lb[k]), None)
        print("first differing position", k, la[max(0, (k or 0) - 2):(k or 0) + 3], lb[max(0, (k or 0) - 2):(k or 0) + 3])
        ex = a.get_exclusions(); print("excl pairs", len(ex), "bad in excl:", [int(np.sum((ex == t).any(axis=1))) for t in bad[:5]])
        break
else:
    print("no difference in 80 steps")
