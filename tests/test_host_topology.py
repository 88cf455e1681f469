"""CPU-only test of the product's host topology manager (chemlab_amd/csrc/chem_host.hpp: bond graph, angle
spawning for registered type triples, exclusions, cluster labels, bonded/exclusion CSR) against an independent
Python model of the rules in SURVEY 3.4 / DESIGN 3 (reference: TopologyManager wiring,
/root/reference/src/start_simulation.py:378-441).  The harness is compiled with g++ from tests/host/."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("host") / "topology_harness")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "topology_harness.cpp"), "-o", exe])
    return exe


def model(n, types, res, init_bonds, reg, batches):
    """Independent restatement: returns (bonds, angles, excl rows, labels)."""
    graph = [set() for _ in range(n)]
    excl = [set() for _ in range(n)]
    bonds, angles, seen_b, seen_a = [], [], set(), set()
    res = list(res); mol = list(range(n))

    def key(t):
        return min(tuple(t), tuple(reversed(t)))

    def add_bond(a, b):
        if key((a, b)) in seen_b:
            return False
        seen_b.add(key((a, b))); bonds.append((a, b)); return True
    for a, b in init_bonds:
        if add_bond(a, b):
            graph[a].add(b); graph[b].add(a); excl[a].add(b); excl[b].add(a)
    for batch in batches:
        new = [(a, b) for a, b in batch if add_bond(a, b)]
        for a, b in new:
            graph[a].add(b); graph[b].add(a)
        for a, b in new:                                   # labels: flood the merged cluster, in event order
            nr, nm = min(res[a], res[b]), min(mol[a], mol[b])
            stack, seen = [a], {a}
            while stack:
                p = stack.pop(); res[p] = nr; mol[p] = nm
                for q in graph[p]:
                    if q not in seen:
                        seen.add(q); stack.append(q)
        for a, b in new:
            excl[a].add(b); excl[b].add(a)
            cands = [(x, a, b) for x in sorted(graph[a]) if x != b] + [(a, b, m) for m in sorted(graph[b]) if m != a]
            for t in cands:
                ty = tuple(types[x] for x in t)
                for r in reg:
                    fwd, rev = ty == tuple(r), tuple(reversed(ty)) == tuple(r)
                    if not (fwd or rev):
                        continue
                    tt = t if fwd else tuple(reversed(t))
                    if key(tt) not in seen_a:
                        seen_a.add(key(tt)); angles.append(tt)
                        excl[tt[0]].add(tt[2]); excl[tt[2]].add(tt[0])
                    break
    return bonds, angles, [sorted(e) for e in excl], list(zip(res, mol))


def run_harness(exe, n, types, res, init_bonds, reg, batches, reserve_after=None):
    lines = ["n %d" % n] + ["type %d %d" % (i, t) for i, t in enumerate(types)] + ["res %d %d" % (i, r) for i, r in enumerate(res)]
    lines += ["list 2", "list 3"] + ["reg 1 %d %d %d" % tuple(r) for r in reg]
    lines += ["bond 0 %d %d" % b for b in init_bonds]
    for k, batch in enumerate(batches):
        if reserve_after is not None and k == reserve_after:
            lines.append("reserve 0 %d" % (50 * n))           # re-hash of a populated set into a table sized for a whole run
        lines.append("newbonds %d " % len(batch) + " ".join("%d %d" % b for b in batch))
    lines.append("dump")
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
    it = iter(out)
    lists = {}
    line = next(it)
    while line.startswith("list"):
        _, li, ar, cnt = line.split()
        lists[int(li)] = [tuple(int(x) for x in next(it).split()) for _ in range(int(cnt))]
        line = next(it)
    assert line.startswith("excl")
    npairs = int(line.split()[1])
    excl = [[int(x) for x in next(it).split(":")[1].split()] for _ in range(n)]
    assert next(it) == "labels"
    labels = [tuple(int(x) for x in next(it).split()) for _ in range(n)]
    nent = int(next(it).split()[1])
    csr = [next(it).split(":", 1)[1] for _ in range(n)]
    return lists, npairs, excl, labels, nent, csr


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_topology_manager_against_python_model(harness, seed):
    rng = np.random.default_rng(seed)
    n = 60
    types = rng.integers(0, 3, n).tolist()
    res = (np.arange(n) // 3 + 1).tolist()
    init = [(3 * k, 3 * k + 1) for k in range(n // 3)]            # one bond per residue to start with
    reg = [(0, 1, 2), (1, 1, 1), (0, 0, 1), (2, 0, 2)]
    batches = []
    for _ in range(4):                                            # disjoint pairs per batch, like one reaction step
        perm = rng.permutation(n)[:16]
        batches.append([(int(perm[2 * k]), int(perm[2 * k + 1])) for k in range(8)])
    lists, npairs, excl, labels, nent, csr = run_harness(harness, n, types, res, init, reg, batches)
    bonds, angles, mexcl, mlabels = model(n, types, res, init, reg, batches)
    assert lists[0] == bonds
    assert lists[1] == angles
    assert excl == mexcl and npairs == sum(len(e) for e in mexcl) // 2
    assert labels == mlabels
    # bonded CSR: one entry per tuple member, in list order, member position recorded
    assert nent == 2 * len(bonds) + 3 * len(angles)
    for t, row in enumerate(csr):
        want = ["(%d %d 0 s0 m%d)" % (a, b, (a, b).index(t)) for a, b in bonds if t in (a, b)]
        want += ["(%d %d %d s1 m%d)" % (a, b, c, (a, b, c).index(t)) for a, b, c in angles if t in (a, b, c)]
        assert row.replace(") (", ")|(").strip().split("|") == want if want else row.strip() == ""


@pytest.mark.parametrize("reserve_after", [0, 2])
def test_reserving_the_bond_hash_set_changes_nothing(harness, reserve_after):
    """chem_run sizes the de-duplication set of the reaction bond list for one bond per particle before the first
    reaction step (a re-hash per doubling cost milliseconds mid-run): same lists, exclusions, labels and CSR whether the
    set is re-hashed empty, populated, or never."""
    rng = np.random.default_rng(7)
    n = 90
    types = rng.integers(0, 3, n).tolist()
    res = (np.arange(n) // 3 + 1).tolist()
    init = [(3 * k, 3 * k + 1) for k in range(n // 3)]
    reg = [(0, 1, 2), (1, 1, 1)]
    batches = []
    for _ in range(5):
        perm = rng.permutation(n)[:24]
        batches.append([(int(perm[2 * k]), int(perm[2 * k + 1])) for k in range(12)])
    batches.append(batches[1][:6] + [(b, a) for a, b in batches[2][:6]])      # duplicates, also reversed: must be rejected either way
    ref = run_harness(harness, n, types, res, init, reg, batches)
    got = run_harness(harness, n, types, res, init, reg, batches, reserve_after=reserve_after)
    assert got == ref
