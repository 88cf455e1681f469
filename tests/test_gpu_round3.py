"""Round-3 GPU parity tests (HIP path through the C ABI vs the CPU oracle): thermal groups, ATRPActivator, the dense
per-cell fall-back, fp32 bonded terms, and the headline size (C5: 1M particles) against the oracle."""
import os

import numpy as np
import pytest

from chemlab_amd import workloads as W
from conftest import rel_err
from helpers import force_error_without_cutoff_flips, sorted_events
from test_gpu_parity import TOL, both

pytestmark = pytest.mark.gpu

# fp32 pair forces at the headline size (box edge 107.7) with the cutoff-boundary decisions set apart: positions are int32
# fixed point (5e-8), staged tile-local (|u| <= 7.1, rounding 2.4e-7), pair arithmetic in fp32
TOL32_C5 = 2e-5


def test_thermal_groups_trajectory_matches_oracle(make_gpu, make_oracle):
    """LangevinThermostat.add_valid_types (start_simulation.py:312-336): friction and noise only on the listed types, with
    types that change through reactions (B -> D and back), fp64 trajectory against the oracle."""
    spec = W.reactive_melt(n=8788, seed=41, interval=10)
    spec["thermal_types"] = [0, 2]                   # A and D thermalised, B not
    g, o, h = both(make_gpu, make_oracle, spec, 64)
    for _ in range(3):
        g.run(20); o.run(20)
    assert len(o.get_events()) > 500
    assert [e[:4] for e in sorted_events(g.get_events())] == [e[:4] for e in sorted_events(o.get_events())]
    assert rel_err(g.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < 1e-8
    assert rel_err(g.get_state("VEL"), o.get_state("VEL")) < 1e-7
    # and it is not a no-op: the same run with every type thermalised ends elsewhere
    g2 = make_gpu(64)
    W.apply(dict(spec, thermal_types=[]), g2)
    for _ in range(3):
        g2.run(20)
    assert rel_err(g2.get_state("VEL"), o.get_state("VEL")) > 1e-3


@pytest.mark.parametrize("prec", [64, 32])
def test_atrp_activator_matches_oracle(make_gpu, make_oracle, prec):
    """ATRPActivator (reaction_post_process.py:380-426): dormant A centres (state 0) are activated into the state window of
    the chain-growth reactions, active ends are deactivated again; flips, catalyst fractions, the reactions they enable and
    the state/type vectors must equal the oracle's."""
    spec = W.reactive_melt(n=8788, seed=43, interval=20)
    spec["state"] = np.where(spec["types"] == 0, 0, 1).astype(np.int32)      # every A dormant
    if prec == 32:          # frozen fp32-representable positions: identical discrete outcomes are then required in fp32 too
        spec["box"] = [float(np.float32(spec["box"][0]))] * 3
        spec = W.snap_to_grid(spec)    # positions both builds represent exactly (fp32 build: int32 fixed point)
        spec["dt"] = 1e-9
        spec["vel"] = np.zeros_like(spec["vel"])
        for r in spec["reaction"]["reactions"]:
            r["rate"] = 1e12                          # (acceptance probability rate * dt * interval must not vanish with dt)
    spec["atrp"] = dict(interval=10, num_particles=1500, ratio_activator=0.5, ratio_deactivator=0.5, delta_catalyst=0.3,
                        k_activate=1.0, k_deactivate=0.6, select_from_all=True, seed=17,
                        centers=[dict(type_id=0, state=0, is_activator=False, new_type=0, new_mass=1.0, delta_state=1),
                                 dict(type_id=0, state=3, is_activator=True, new_type=0, new_mass=1.0, delta_state=-3)])
    g, o, h = both(make_gpu, make_oracle, spec, prec, thermostat=(prec == 64))
    for _ in range(4):
        g.run(25); o.run(25)                         # firings at 10, 20, ... interleaved with reaction steps at 20, 40, ...
    sg, so = g.atrp_stats(), o.atrp_stats()
    assert len(so) == 10 and sum(r["activated"] for r in so) > 200
    assert sg == so
    assert len(o.get_events()) > 50
    assert [e[:4] for e in sorted_events(g.get_events())] == [e[:4] for e in sorted_events(o.get_events())]
    assert np.array_equal(g.get_state("STATE"), o.get_state("STATE"))
    assert np.array_equal(g.get_state("TYPE"), o.get_state("TYPE"))
    assert np.array_equal(g.get_list(h["reaction_bonds"]), o.get_list(h["reaction_bonds"]))


@pytest.mark.parametrize("prec", [64, 32])
def test_dense_system_leaves_the_lds_tiles_and_stays_correct(make_gpu, make_oracle, prec):
    """A system whose 125-cell stencil (~8000 particles) fits neither the force kernel's nor the list build's LDS image:
    the engine must fall back to the per-cell kernels (whose 27-cell stencil of ~1700 particles is walked in windows of
    1536 slots) instead of failing, and agree with the oracle."""
    rng = np.random.default_rng(12)
    L, n = 14.2, 8000
    k = 20
    gpts = (np.stack(np.meshgrid(np.arange(k), np.arange(k), np.arange(k), indexing="ij"), -1).reshape(-1, 3) + 0.5) * (L / k)
    pos = gpts + rng.uniform(-0.1, 0.1, gpts.shape)
    spec = dict(n=n, box=[L] * 3, rc=2.5, skin=0.3, dt=0.002, ids=np.arange(1, n + 1), types=(np.arange(n) % 2).astype(np.int32), pos=pos,
                vel=rng.normal(0, 0.3, (n, 3)), mass=np.ones(n), lj=[(0, 0, 0.5, 0.45, 2.5), (0, 1, 0.5, 0.45, 2.5), (1, 1, 0.5, 0.45, 2.5)],
                kT=1.0, gamma=0.0, seed=3, exclusions=np.stack([np.arange(1, 2001, 2), np.arange(2, 2002, 2)], 1))
    g, o, _ = both(make_gpu, make_oracle, spec, prec, thermostat=False)
    g.run(0); o.run(0)
    # (soft system: the largest force is ~1, so the rounding of the staged coordinates weighs more against it than in the LJ melts)
    assert rel_err(g.get_state("FORCE"), o.get_state("FORCE")) < (TOL[64] if prec == 64 else 2e-5)
    if prec == 64:
        assert np.array_equal(g.get_verlet_pairs(), o.get_verlet_pairs())
    g.run(20); o.run(20)
    assert rel_err(g.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < (1e-9 if prec == 64 else 1e-4)


def test_bonded_terms_fp32_against_oracle(make_gpu, make_oracle):
    """FENE bonds, cosine angles, n-cosine / Ryckaert-Bellemans / tabulated dihedrals in production precision (fp32 arrays,
    fp64 bonded geometry): forces within the fp32 pair tolerance of the largest force, list energies to 1e-5."""
    rng = np.random.default_rng(8)
    nm = 125
    base = (np.stack(np.meshgrid(np.arange(5), np.arange(5), np.arange(5), indexing="ij"), -1).reshape(-1, 3) * 4.0 + 1.0)
    off = np.array([[0, 0, 0], [0.9, 0.2, 0.1], [1.2, 1.1, 0.4], [2.1, 1.3, 1.2]])
    pos = (base[:, None, :] + off[None]).reshape(-1, 3) + rng.uniform(-0.05, 0.05, (4 * nm, 3))
    n = 4 * nm
    ids = np.arange(1, n + 1).reshape(nm, 4)
    spec = dict(n=n, box=[20.0] * 3, rc=2.5, skin=0.3, dt=0.002, ids=np.arange(1, n + 1), types=np.zeros(n, np.int32),
                pos=pos, vel=rng.normal(0, 0.3, (n, 3)), mass=np.ones(n), lj=[(0, 0, 0.2, 0.8, 2.5)], kT=1.0, gamma=0.0, seed=1,
                lists=[dict(arity=2, kind="FENE", params=[30.0, 0.0, 2.5], ids=np.concatenate([ids[:, [0, 1]], ids[:, [1, 2]], ids[:, [2, 3]]])),
                       dict(arity=3, kind="ANG_COSINE", params=[2.0, np.deg2rad(130)], ids=np.concatenate([ids[:, [0, 1, 2]], ids[:, [1, 2, 3]]])),
                       dict(arity=4, kind="DIH_NCOS", params=[1.5, np.deg2rad(20), 3.0], ids=ids),
                       dict(arity=4, kind="DIH_RB", params=[0.5, -0.3, 0.2, 0.1, -0.1, 0.05], ids=ids[::2])],
                exclusions=np.concatenate([ids[:, [0, 1]], ids[:, [1, 2]], ids[:, [2, 3]], ids[:, [0, 2]], ids[:, [1, 3]], ids[:, [0, 3]]]))
    g, o, _ = both(make_gpu, make_oracle, spec, 32)
    dphi = 2 * np.pi / 720
    phi = -np.pi + dphi * np.arange(721)
    for eng in (g, o):
        h = eng.list_create(4, "DIH_TABULATED")
        eng.list_set_params(h, [eng.table_create(phi[0], dphi, 0.8 * (1 + np.cos(2 * phi - 0.3)), 1.6 * np.sin(2 * phi - 0.3))])
        eng.list_add(h, ids[::3])
    g.run(0); o.run(0)
    assert rel_err(g.get_state("FORCE"), o.get_state("FORCE")) < TOL[32]
    og, oo = g.observe(), o.observe()
    assert np.allclose(og["epot_list"], oo["epot_list"], rtol=1e-5)
    g.run(50); o.run(50)
    assert rel_err(g.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < 1e-4


def test_topology_manager_trimer_melt_fp32(make_gpu, make_oracle):
    """test_topology_manager_parity_trimer_melt in production precision on frozen positions: spawned angles, exclusions,
    residue labels identical; bonded energies to fp32 accuracy."""
    spec = W.trimer_melt(n_mol=216, seed=4, interval=1)
    spec["reaction"]["reactions"][0]["cutoff"] = 1.4           # the facing MA ends start 1.26 apart: reactive without moving
    spec["box"] = [float(np.float32(b)) for b in spec["box"]]
    spec = W.snap_to_grid(spec)        # positions both builds represent exactly (fp32 build: int32 fixed point)
    spec["dt"] = 1e-9
    spec["vel"] = np.zeros_like(spec["vel"])
    g, o, h = both(make_gpu, make_oracle, spec, 32, thermostat=False)
    g.run(2); o.run(2)
    assert len(o.get_events()) > 5
    assert [e[:4] for e in sorted_events(g.get_events())] == [e[:4] for e in sorted_events(o.get_events())]
    for k in (0, 1, "reaction_bonds"):
        assert np.array_equal(g.get_list(h[k]), o.get_list(h[k]))
    assert np.array_equal(g.get_exclusions(), o.get_exclusions())
    assert np.array_equal(g.get_state("RESID"), o.get_state("RESID"))
    og, oo = g.observe(), o.observe()
    assert np.allclose(og["epot_list"], oo["epot_list"], rtol=1e-5, atol=1e-4)
    assert og["list_size"] == oo["list_size"]
    assert rel_err(g.get_state("FORCE"), o.get_state("FORCE")) < TOL[32]


def test_headline_c5_one_million_particles_against_the_oracle(make_gpu, make_oracle):
    """BASELINE C5 (the bench workload): 1M-particle reactive LJ melt.  The lattice is melted on the GPU (1500 fp32 steps
    without reactions), then ONE force evaluation of that configuration in fp32 and fp64 is compared with the oracle
    (forces, epot_lj, virial), and one reaction step on the frozen (fp32-representable) configuration must give the
    oracle's events, bonds, states and types bit for bit."""
    spec = W.reactive_melt(n=1000000, rho=0.8, seed=2, interval=1)
    spec["box"] = [float(np.float32(spec["box"][0]))] * 3
    m = make_gpu(32)
    W.apply(spec, m, reactions=False)
    m.run(1500)
    pos = m.get_state("POS")                                   # decoded fixed-point coordinates: exact in both builds and for the oracle
    x0 = np.asarray(spec["pos"])
    d = m.get_state("POS_UNFOLDED") - x0
    assert (d * d).sum(1).mean() > 0.5                          # the lattice is gone
    m.close()
    spec["pos"] = pos
    spec["vel"] = np.zeros_like(spec["vel"])
    spec["dt"] = 1e-9
    for r in spec["reaction"]["reactions"]:
        r["rate"] = 1e12
    o = make_oracle()
    ho = W.apply(spec, o, thermostat=False)
    o.run(0)
    fo, oo = o.get_state("FORCE"), o.observe()
    for prec in (64, 32):
        g = make_gpu(prec)
        hg = W.apply(spec, g, thermostat=False)
        g.run(0)
        fg = g.get_state("FORCE")
        err = rel_err(fg, fo)
        og = g.observe()
        msg = "C5 1M force parity, fp%d: max |dF| / max |F| = %.3e, epot_lj rel %.3e" % (prec, err, abs(og["epot_lj"] / oo["epot_lj"] - 1))
        if prec == 64:
            assert err < TOL[64], msg
        else:
            # raw maximum: a handful of pairs within rounding of rc, each worth |F(rc)| = 0.039 = 2.6e-4 of the largest force;
            # with those boundary decisions set apart the error is the arithmetic's
            err2, flips = force_error_without_cutoff_flips(spec, fg, fo, TOL32_C5, max_flips=100)
            msg += "; without %d cutoff-boundary pairs: %.3e" % (flips, err2)
            assert err < 5e-4 and err2 < TOL32_C5 and 0 <= flips <= 100, msg
        print(msg)
        assert og["epot_lj"] == pytest.approx(oo["epot_lj"], rel=1e-11 if prec == 64 else 2e-6)
        assert og["virial_nb"] == pytest.approx(oo["virial_nb"], rel=1e-10 if prec == 64 else 1e-5)
        if prec == 32:
            g.run(1)
            if not len(o.get_events()):
                o.run(1)
            eg, eo = sorted_events(g.get_events()), sorted_events(o.get_events())
            assert len(eo) > 100000 and [e[:4] for e in eg] == [e[:4] for e in eo]
            assert np.allclose([e[4] for e in eg], [e[4] for e in eo], rtol=1e-6)
            assert np.array_equal(g.get_state("STATE"), o.get_state("STATE"))
            assert np.array_equal(g.get_state("TYPE"), o.get_state("TYPE"))
            assert np.array_equal(g.get_list(hg["reaction_bonds"]), o.get_list(ho["reaction_bonds"]))
        g.close()


def test_checkup_invariant_of_chain_growth_catalytic(tmp_path, monkeypatch):
    """The one output the reference holds for this path: examples/chain_growth_catalytic/Checkup.ipynb builds the graph of
    the A-A bonds of a finished run and prints the set of node degrees, stored cell output `{0, 1, 2, 3}` -- no A bead
    ever carries more than three chain bonds.  The shipped example through the py3 driver on the GPU (production precision)."""
    import shutil
    from chemlab_amd import start_simulation
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chain_growth_catalytic")
    d = tmp_path / "run"
    shutil.copytree(gold, str(d))
    monkeypatch.chdir(d)
    res = start_simulation.main(["@params"], quiet=True)          # the shipped parameters: run=5000, reactions from step 2000 every 500
    e = res["system"].engine
    bonds = np.asarray(res["chem_fpls"][0][1].getAllBonds(), dtype=np.int64).reshape(-1, 2)
    assert len(bonds) > 50
    deg = np.bincount(bonds.ravel(), minlength=e.n + 1)
    assert set(np.unique(deg).tolist()) <= {0, 1, 2, 3}          # the notebook's stored output
    assert deg.max() >= 2                                        # chains did grow beyond dimers
    # the same from the trajectory file, the way the notebook does it (last frame of /connectivity/chem_bonds_0)
    if res["trajectory"].endswith(".npz"):
        z = np.load(res["trajectory"])
        last = z["connectivity/chem_bonds_0/value"][-1]
        real = last[(last != -1).all(axis=1)]
        assert np.array_equal(np.sort(real, axis=0), np.sort(bonds, axis=0))
    ty = e.get_state("TYPE")
    ids = e.get_state("ID")
    a_type = res["gt"].atomsym_atomtype["A"]
    bonded_ids = np.unique(bonds.ravel())
    assert np.all(ty[np.searchsorted(ids, bonded_ids)] == a_type)   # the chain bonds join A beads only (reaction.cfg b, c)


@pytest.mark.parametrize("prec", [64, 32])
def test_wider_internal_list_skin_changes_nothing_but_the_rebuild_rate(make_gpu, make_oracle, prec):
    """Option list_skin: cells, tiles and the 16-bit force list are built for rc + list_skin > rc + skin and live until the
    accumulated displacement exceeds list_skin / 2.  The force kernel applies the exact cutoff, so forces, events and
    trajectories are those of the workload's skin; the reported rebuild count stays the reference rule's, the int32 Verlet
    rows stay the pairs within rc + skin."""
    spec = W.reactive_melt(n=8788, seed=51, interval=25)
    spec["rebuild_criterion"] = 0
    g, o, h = both(make_gpu, make_oracle, spec, prec)
    g.set_option("list_skin", 0.6)
    g.run(0); o.run(0)
    assert rel_err(g.get_state("FORCE"), o.get_state("FORCE")) < TOL[prec]
    for _ in range(3):
        g.run(25); o.run(25)
    tg, to = g.timers(), o.timers()
    assert tg["rebuilds"] == to["rebuilds"] >= 8                     # the reference rule's count (rebuilds forced by reaction steps included)
    assert 3 <= tg["list_rebuilds"] < tg["rebuilds"]                 # the lists themselves were built less often
    if prec == 64:
        assert [e[:4] for e in sorted_events(g.get_events())] == [e[:4] for e in sorted_events(o.get_events())]
        assert np.array_equal(g.get_list(h["reaction_bonds"]), o.get_list(h["reaction_bonds"]))
        assert rel_err(g.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < 1e-8
    else:
        assert rel_err(g.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < 1e-4
    if prec == 64:       # (builds the int32 rows on demand: a forced rebuild, hence behind the count comparison)
        assert np.array_equal(g.get_verlet_pairs(), o.get_verlet_pairs())
    # against the same engine without the option: identical up to summation order
    g2 = make_gpu(prec)
    W.apply(spec, g2)
    g2.set_option("list_skin", 0)
    g2.run(0)
    for _ in range(3):
        g2.run(25)
    assert g2.timers()["list_rebuilds"] == g2.timers()["rebuilds"] == to["rebuilds"]
    assert rel_err(g.get_state("POS_UNFOLDED"), g2.get_state("POS_UNFOLDED")) < (1e-9 if prec == 64 else 1e-4)


def test_restrict_reaction_matches_oracle(make_gpu, make_oracle):
    """RestrictReaction.define_connection (reaction_setup.py:75-78,115-128): the candidate filter by connectivity map on the
    device (per-tag CSR of allowed partners) against the oracle's, event log and bonds bit-identical (fp64)."""
    spec = W.reactive_melt(n=8788, seed=61, interval=10)
    for r in spec["reaction"]["reactions"]:
        r["rate"] = 1e9
    probe = make_oracle()
    W.apply(spec, probe)
    probe.run(30)
    ev = probe.get_events()
    allowed = np.stack([ev["id_a"], ev["id_b"]], 1)[ev["reaction"] == 1][::2]
    assert len(allowed) > 50
    g, o, h = both(make_gpu, make_oracle, spec, 64)
    g.reaction_restrict(1, allowed); o.reaction_restrict(1, allowed)
    g.run(30); o.run(30)
    eg, eo = sorted_events(g.get_events()), sorted_events(o.get_events())
    assert [e[:4] for e in eg] == [e[:4] for e in eo] and len(eo) > 100
    got = {tuple(sorted(e[1:3])) for e in eo if e[3] == 1}
    assert got and got <= {tuple(sorted(p)) for p in allowed.tolist()}
    assert np.array_equal(g.get_list(h["reaction_bonds"]), o.get_list(h["reaction_bonds"]))
    assert np.array_equal(g.get_state("STATE"), o.get_state("STATE"))


def test_exchange_reaction_matches_oracle(make_gpu, make_oracle):
    """`A:B + C -> A:C + B` as the reference sets it up (reaction_setup.py:167-251): virtual A + C reaction, neighbour-state
    constraint on A (bit table from the host's bond graph), the B neighbour retyped with its state incremented inside a
    state window -- events, types and states against the oracle (fp64)."""
    from test_oracle_extensions import _apply_exchange, _exchange_spec
    spec = _exchange_spec(n_mol=1500, seed=33)
    spec.setdefault("rebuild_criterion", 1)
    g, o = make_gpu(64), make_oracle()
    _apply_exchange(spec, g); _apply_exchange(spec, o)
    for _ in range(3):
        g.run(5); o.run(5)
    eg, eo = sorted_events(g.get_events()), sorted_events(o.get_events())
    assert len(eo) > 100 and [e[:4] for e in eg] == [e[:4] for e in eo]
    assert np.array_equal(g.get_state("TYPE"), o.get_state("TYPE"))
    assert np.array_equal(g.get_state("STATE"), o.get_state("STATE"))
    assert (o.get_state("TYPE") == 3).sum() == len(eo)
    assert rel_err(g.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < 1e-8


@pytest.mark.parametrize("prec", [64, 32])
def test_lj_14_pair_list_matches_oracle(make_gpu, make_oracle, prec):
    """FixedPairListLennardJones (1-4 pairs): forces and the shifted list energy against the oracle."""
    rng = np.random.default_rng(4)
    n = 600
    k = 9
    pts = (np.stack(np.meshgrid(np.arange(k), np.arange(k), np.arange(k), indexing="ij"), -1).reshape(-1, 3)[:n] + 0.5) * 1.3
    pos = pts + rng.uniform(-0.08, 0.08, pts.shape)
    ids = np.arange(1, n + 1)
    pairs = np.stack([ids[0:n - 1:2], ids[1:n:2]], 1)
    spec = dict(n=n, box=[k * 1.3] * 3, rc=2.5, skin=0.3, dt=0.002, ids=ids, types=np.zeros(n, np.int32), pos=pos, vel=np.zeros((n, 3)),
                mass=np.ones(n), lj=[(0, 0, 1.0, 1.0, 2.5)], kT=1.0, gamma=0.0, seed=1,
                lists=[dict(arity=2, kind="LJ_BOND", params=[0.5, 1.05, 2.2], ids=pairs)], exclusions=pairs)
    g, o, h = both(make_gpu, make_oracle, spec, prec, thermostat=False)
    g.run(0); o.run(0)
    assert rel_err(g.get_state("FORCE"), o.get_state("FORCE")) < TOL[prec]
    og, oo = g.observe(), o.observe()
    assert og["epot_list"][h[0]] == pytest.approx(oo["epot_list"][h[0]], rel=1e-11 if prec == 64 else 1e-5)
    assert oo["epot_list"][h[0]] != 0.0


@pytest.mark.parametrize("prec", [64, 32])
def test_run_stopped_by_the_device_recovers_and_changes_nothing(make_gpu, prec):
    """A cell that outgrows its bucket row in the middle of an asynchronous run (fused rebuild): the device stops the run at
    that step (every later launch leaves at once), the host finds out at its next synchronisation, widens the rows and
    resumes there.  Provoked with rows of 33 slots (option bucket_cap) on a lattice start whose fullest cell holds 32: the
    first builds fit, the melting system overflows a few dozen steps in.  The trajectory -- Langevin noise, reactions, bonds
    -- must be bit-identical to an engine that never had to stop."""
    spec = W.reactive_melt(n=8788, seed=71, interval=40)
    spec["rebuild_criterion"] = 0
    a, b = make_gpu(prec), make_gpu(prec)
    ha, hb = W.apply(spec, a), W.apply(spec, b)
    a.run(0)
    tiles = a.get_state("POS")          # (initial occupancy: the same lattice for both)
    L = spec["box"][0]; nc = int(L // 2.8)
    occ = np.bincount(np.ravel_multi_index(np.minimum((tiles / (L / nc)).astype(int), nc - 1).T, (nc, nc, nc)), minlength=nc ** 3).max()
    b.set_option("bucket_cap", int(occ) + 1)
    a.run(120); b.run(120)
    assert b.timers()["rebuilds"] == a.timers()["rebuilds"]
    assert np.array_equal(a.get_state("POS_UNFOLDED"), b.get_state("POS_UNFOLDED"))
    assert np.array_equal(a.get_state("VEL"), b.get_state("VEL"))
    assert sorted_events(a.get_events()) == sorted_events(b.get_events()) and len(a.get_events()) > 100
    assert np.array_equal(a.get_list(ha["reaction_bonds"]), b.get_list(hb["reaction_bonds"]))
    lib = b.api.lib
    lib.chem_debug_halts.restype = __import__("ctypes").c_int64
    assert lib.chem_debug_halts(__import__("ctypes").c_void_p(b.ctx)) >= 1      # it did happen


@pytest.mark.parametrize("prec", [64, 32])
def test_inline_bonds_equal_the_bonded_kernel(make_gpu, make_oracle, prec):
    """Harmonic bonds evaluated in the force kernel's epilogue from its staged image (partner LDS slots recorded by the list
    build; no bonded launch) against the per-step bonded kernel and the oracle, on a reacting system whose bonds, exclusions
    and list rebuilds keep changing."""
    spec = W.reactive_melt(n=8788, seed=81, interval=10)
    for r in spec["reaction"]["reactions"]:
        r["rate"] = 1e9
    a, b, o = make_gpu(prec), make_gpu(prec), make_oracle()
    ha, hb, ho = W.apply(spec, a), W.apply(spec, b), W.apply(spec, o)
    b.set_option("bonds_inline", 0)
    for _ in range(4):
        a.run(10); b.run(10); o.run(10)
    assert len(o.get_events()) > 2000 and len(o.get_list(ho["reaction_bonds"])) > 300
    fa, fb, fo = a.get_state("FORCE"), b.get_state("FORCE"), o.get_state("FORCE")
    if prec == 64:
        assert [e[:4] for e in sorted_events(a.get_events())] == [e[:4] for e in sorted_events(o.get_events())]
        assert np.array_equal(a.get_list(ha["reaction_bonds"]), o.get_list(ho["reaction_bonds"]))
        assert rel_err(a.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < 1e-8
        assert rel_err(a.get_state("POS_UNFOLDED"), b.get_state("POS_UNFOLDED")) < 1e-9
    else:
        assert rel_err(a.get_state("POS_UNFOLDED"), b.get_state("POS_UNFOLDED")) < 1e-4
    # a static comparison on one and the same configuration (the trajectories above have drifted apart by rounding)
    spec2 = dict(spec, pos=a.get_state("POS"), vel=a.get_state("VEL"), types=a.get_state("TYPE"), state=a.get_state("STATE"))
    spec2["lists"] = [dict(arity=2, kind="HARMONIC", params=[30.0, 0.97], ids=a.get_list(ha["reaction_bonds"]))]
    spec2["exclusions"] = a.get_exclusions()
    # c: the default (the list build ignores the exclusions, a pass behind it records the partner slots, the force kernel
    # takes the partners' pair term out again); c1: the list build removes the excluded pairs and records the slots itself;
    # d: no inline bonds at all
    c, c1, d, o2 = make_gpu(prec), make_gpu(prec), make_gpu(prec), make_oracle()
    for e in (c, c1, d, o2):
        W.apply(spec2, e, thermostat=False, reactions=False)
    d.set_option("bonds_inline", 0)
    c1.set_option("bond_pass", 0)
    c.run(0); c1.run(0); d.run(0); o2.run(0)
    fo2, oo2 = o2.get_state("FORCE"), o2.observe()
    for e in (c, c1, d):
        assert rel_err(e.get_state("FORCE"), fo2) < TOL[prec]
        oe = e.observe()
        assert oe["epot_list"][0] == pytest.approx(oo2["epot_list"][0], rel=1e-11 if prec == 64 else 1e-5)
        assert oe["epot_lj"] == pytest.approx(oo2["epot_lj"], rel=1e-11 if prec == 64 else 2e-6)      # (excluded pairs: not in the energy either)
        assert oe["virial_nb"] == pytest.approx(oo2["virial_nb"], rel=1e-10 if prec == 64 else 2e-5)
    assert rel_err(c.get_state("FORCE"), d.get_state("FORCE")) < (1e-12 if prec == 64 else 1e-5)
    assert rel_err(c1.get_state("FORCE"), d.get_state("FORCE")) < (1e-12 if prec == 64 else 1e-5)
    if prec == 64:
        assert np.array_equal(c.get_verlet_pairs(), o2.get_verlet_pairs())      # (the int32 rows leave the excluded pairs out in any mode)
        c.run(0)
        assert rel_err(c.get_state("FORCE"), fo2) < TOL[prec]                   # ... and the force list is the regular one again behind them


@pytest.mark.parametrize("narm", [6, 10])
@pytest.mark.parametrize("prec", [64, 32])
def test_inline_bonds_with_crowded_centres(make_gpu, make_oracle, prec, narm):
    """Inline bonds beyond the located-partner path: star molecules whose centre carries SIX bonded (= excluded) partners are
    recorded by the list build's generic sweep (second slot quad); with TEN -- more than the eight recorded slots -- the
    centres stay with the work-list kernel while the arms (one bond each) are evaluated inline.  Forces against the oracle
    and against the all-work-list path, then a short trajectory."""
    rng = np.random.default_rng(17)
    k = 9
    cells = (np.stack(np.meshgrid(np.arange(k), np.arange(k), np.arange(k), indexing="ij"), -1).reshape(-1, 3) + 0.5) * 3.3
    arms = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1],
                     [1, 1, 1], [-1, -1, 1], [1, -1, -1], [-1, 1, -1]], float)[:narm]
    arms /= np.linalg.norm(arms, axis=1)[:, None]
    pos = (cells[:, None, :] + np.concatenate([np.zeros((1, 3)), arms])[None]).reshape(-1, 3) + rng.uniform(-0.05, 0.05, (len(cells) * (narm + 1), 3))
    n = len(pos)
    ids = np.arange(1, n + 1).reshape(-1, narm + 1)
    bonds = np.concatenate([np.stack([ids[:, 0], ids[:, a]], 1) for a in range(1, narm + 1)])
    spec = dict(n=n, box=[k * 3.3] * 3, rc=2.5, skin=0.3, dt=0.002, ids=np.arange(1, n + 1), types=np.zeros(n, np.int32), pos=pos,
                vel=rng.normal(0, 0.3, (n, 3)), mass=np.ones(n), lj=[(0, 0, 0.5, 0.9, 2.5)], kT=1.0, gamma=0.0, seed=1,
                lists=[dict(arity=2, kind="HARMONIC", params=[40.0, 1.0], ids=bonds)], exclusions=bonds, rebuild_criterion=0)
    a, b, o = make_gpu(prec), make_gpu(prec), make_oracle()
    for e in (a, b, o):
        W.apply(spec, e, thermostat=False)
    b.set_option("bonds_inline", 0)
    a.run(0); b.run(0); o.run(0)
    fo = o.get_state("FORCE")
    assert rel_err(a.get_state("FORCE"), fo) < TOL[prec] and rel_err(b.get_state("FORCE"), fo) < TOL[prec]
    assert rel_err(a.get_state("FORCE"), b.get_state("FORCE")) < (1e-12 if prec == 64 else 1e-5)   # (fp32: 2 K x the 2.4e-7 of a staged coordinate)
    a.run(60); o.run(60)
    assert a.timers()["rebuilds"] == o.timers()["rebuilds"] >= 2
    assert rel_err(a.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < (1e-9 if prec == 64 else 1e-4)


def _tiles_in_use(g):
    import ctypes
    out = (ctypes.c_int32 * 6)()
    g.api.lib.chem_debug_tiles.restype = ctypes.c_int64
    assert g.api.lib.chem_debug_tiles(ctypes.c_void_p(g.ctx), out) == 0
    return list(out)


@pytest.mark.parametrize("split", [11, 2, 21])
@pytest.mark.parametrize("prec", [64, 32])
def test_narrow_tiles_change_nothing(make_gpu, make_oracle, prec, split):
    """Option tile_split = nb * 10 + w: the first nb tiles of every tile row are three cells wide, the others w cells (the
    small change that fills the last round of a force launch).  Same lists, forces, events and trajectory as the oracle --
    the tiling is invisible -- including bonds evaluated inline and a frozen-position reaction step."""
    spec = W.reactive_melt(n=8788, seed=61, interval=20)       # 7 cells per axis
    for r in spec["reaction"]["reactions"]:
        r["rate"] = 1e9
    g, o, h = both(make_gpu, make_oracle, spec, prec)
    g.set_option("tile_split", split)
    g.run(0); o.run(0)
    ntiles, nx, nwide, w, rows, cap = _tiles_in_use(g)
    assert nx == 7 and nwide == split // 10 and w == split % 10
    assert ntiles == rows * (nwide + -(-(nx - 3 * nwide) // w))
    assert rel_err(g.get_state("FORCE"), o.get_state("FORCE")) < TOL[prec]
    og, oo = g.observe(), o.observe()
    assert og["epot_lj"] == pytest.approx(oo["epot_lj"], rel=1e-11 if prec == 64 else 2e-6)
    for _ in range(3):
        g.run(20); o.run(20)
    assert g.timers()["rebuilds"] == o.timers()["rebuilds"]
    assert len(o.get_events()) > 1000
    assert rel_err(g.get_state("FORCE"), o.get_state("FORCE")) < (TOL[prec] if prec == 64 else 5e-3)   # (fp32: trajectories have drifted by rounding)
    if prec == 64:
        assert [e[:4] for e in sorted_events(g.get_events())] == [e[:4] for e in sorted_events(o.get_events())]
        assert np.array_equal(g.get_list(h["reaction_bonds"]), o.get_list(h["reaction_bonds"]))
        assert rel_err(g.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < 1e-8
        assert np.array_equal(g.get_verlet_pairs(), o.get_verlet_pairs())      # (last: the int32 rows come from a forced rebuild)
    else:
        assert rel_err(g.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < 1e-4


@pytest.mark.parametrize("prec", [64, 32])
def test_decomposed_path_with_list_skin_and_inline_bonds(make_gpu, make_oracle, prec):
    """The slab path (one rank, its own z-neighbour: dd_self) with this round's additions: lists on the internal list skin
    (reported rebuild count = the reference rule's), harmonic reaction bonds evaluated inline by the force kernel.  Events,
    bonds, rebuild count and trajectory against the oracle, then the forces of the final configuration against the oracle
    and against the same engine with both switched off."""
    spec = W.reactive_melt(n=8788, seed=91, interval=10)
    spec["rebuild_criterion"] = 0
    for r in spec["reaction"]["reactions"]:
        r["rate"] = 1e9
    g, o, s1 = make_gpu(prec), make_oracle(), make_gpu(prec)
    g.set_option("dd_self", 1)
    g.set_option("list_skin", 0.55)
    h = W.apply(spec, g); W.apply(spec, o); W.apply(spec, s1)
    for _ in range(4):
        g.run(10); o.run(10); s1.run(10)
    tg, to, ts = g.timers(), o.timers(), s1.timers()
    # (the first reaction step changes types without making a bond: the HIP engines refresh their type-filtered force list
    #  there, the oracle's list holds all pairs and stays -- one forced build more than the oracle on either HIP path)
    assert tg["rebuilds"] == ts["rebuilds"] and to["rebuilds"] <= tg["rebuilds"] <= to["rebuilds"] + 1
    assert tg["list_rebuilds"] < tg["rebuilds"]
    assert len(o.get_list(h["reaction_bonds"])) > 300
    if prec == 64:
        assert [e[:4] for e in sorted_events(g.get_events())] == [e[:4] for e in sorted_events(o.get_events())]
        assert np.array_equal(g.get_list(h["reaction_bonds"]), o.get_list(h["reaction_bonds"]))
        assert rel_err(g.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < 1e-8
    else:
        assert rel_err(g.get_state("POS_UNFOLDED"), o.get_state("POS_UNFOLDED")) < 1e-4
    spec2 = dict(spec, pos=g.get_state("POS"), vel=g.get_state("VEL"), types=g.get_state("TYPE"), state=g.get_state("STATE"))
    spec2["lists"] = [dict(arity=2, kind="HARMONIC", params=[30.0, 0.97], ids=g.get_list(h["reaction_bonds"]))]
    spec2["exclusions"] = g.get_exclusions()
    c, d, o2 = make_gpu(prec), make_gpu(prec), make_oracle()
    for e in (c, d):
        e.set_option("dd_self", 1)
    d.set_option("bonds_inline", 0); d.set_option("list_skin", 0)
    c.set_option("list_skin", 0.55)
    for e in (c, d, o2):
        W.apply(spec2, e, thermostat=False, reactions=False)
    c.run(0); d.run(0); o2.run(0)
    fo2 = o2.get_state("FORCE")
    assert rel_err(c.get_state("FORCE"), fo2) < TOL[prec] and rel_err(d.get_state("FORCE"), fo2) < TOL[prec]
    assert c.observe()["epot_list"][0] == pytest.approx(o2.observe()["epot_list"][0], rel=1e-11 if prec == 64 else 1e-5)


def test_reaction_scan_without_room_for_the_role_words(make_gpu, make_oracle):
    """fp64, five crowded cells per axis: the 32-byte-per-slot image fills the LDS budget and leaves no room for the staged role
    words of the reaction scan -- it reads them from global memory then (found by a randomised sweep: the decomposed path refused
    this system outright while the role words counted towards the tile budget).  Events and bonds against the oracle, on the
    slab path and on the single-domain path."""
    spec = W.reactive_melt(n=4000, seed=12, interval=7)
    spec["rebuild_criterion"] = 0
    for r in spec["reaction"]["reactions"]:
        r["rate"] = 1e9
    for dd in (1, 0):
        g, o = make_gpu(64), make_oracle()
        g.set_option("dd_self", dd)
        h = W.apply(spec, g); W.apply(spec, o)
        for _ in range(4):
            g.run(7); o.run(7)
        assert len(o.get_events()) > 1000
        assert [e[:4] for e in sorted_events(g.get_events())] == [e[:4] for e in sorted_events(o.get_events())]
        assert np.array_equal(g.get_list(h["reaction_bonds"]), o.get_list(h["reaction_bonds"]))
        assert _tiles_in_use(g)[5] >= 4500                      # (the tile capacity that crowds the LDS)
