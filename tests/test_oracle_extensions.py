"""Oracle (CPU) checks of the round-3 extensions: thermal groups of the Langevin thermostat
(/root/reference/src/start_simulation.py:312-336) and the ATRPActivator rule set
(/root/reference/src/chemlab/reaction_post_process.py:380-426, examples/atrp_lj/atrp.cfg:15-25; include/chem_mi355.h)."""
import os

import numpy as np

from chemlab_amd import workloads as W

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_thermal_groups_thermalise_only_listed_types(make_oracle):
    """Two non-interacting ideal gases (no potential at all): the listed type relaxes to kT, the other keeps its
    velocities exactly (ballistic)."""
    rng = np.random.default_rng(3)
    n = 2000
    spec = dict(n=n, box=[30.0] * 3, rc=2.5, skin=0.3, dt=0.01, ids=np.arange(1, n + 1), types=(np.arange(n) % 2).astype(np.int32),
                pos=rng.uniform(0, 30, (n, 3)), vel=rng.normal(0, 2.0, (n, 3)), mass=np.ones(n), kT=0.5, gamma=5.0, seed=7,
                thermal_types=[1])
    o = make_oracle()
    W.apply(spec, o)
    v0 = np.array(spec["vel"])
    o.run(400)
    v = o.get_state("VEL")
    cold = spec["types"] == 0
    assert np.array_equal(v[cold], v0[cold])                      # no force, no thermostat: untouched
    kT_hot = (v[~cold] ** 2).mean()
    assert abs(kT_hot - 0.5) < 0.06                               # relaxed from kT = 4 to kT = 0.5
    # an empty list means "every type" (the reference passes [] without --thermal_groups)
    o2 = make_oracle()
    W.apply(dict(spec, thermal_types=[]), o2)
    o2.run(400)
    assert abs((o2.get_state("VEL") ** 2).mean() - 0.5) < 0.05


def _atrp_spec(n=3000, k_deact=0.0, select_from_all=True, num=400):
    rng = np.random.default_rng(5)
    DA, FA, X = 0, 1, 2
    types = rng.integers(0, 3, n).astype(np.int32)
    state = np.where(types == X, 0, 2).astype(np.int32)           # dormant centres: state 2
    return dict(n=n, box=[40.0] * 3, rc=2.5, skin=0.3, dt=0.005, ids=np.arange(1, n + 1), types=types, state=state,
                pos=rng.uniform(0, 40, (n, 3)), vel=np.zeros((n, 3)), mass=np.ones(n), kT=1.0, gamma=0.0, seed=9,
                reaction=None,
                atrp=dict(interval=10, num_particles=num, ratio_activator=0.2, ratio_deactivator=0.8, delta_catalyst=0.2,
                          k_activate=1.0, k_deactivate=k_deact, select_from_all=select_from_all, seed=11,
                          centers=[dict(type_id=DA, state=2, is_activator=False, new_type=DA, new_mass=1.0, delta_state=1),
                                   dict(type_id=DA, state=3, is_activator=True, new_type=DA, new_mass=1.0, delta_state=-1),
                                   dict(type_id=FA, state=2, is_activator=False, new_type=FA, new_mass=1.0, delta_state=1),
                                   dict(type_id=FA, state=3, is_activator=True, new_type=FA, new_mass=1.0, delta_state=-1)]))


def test_atrp_activator_catalyst_balance_and_state_flips(make_oracle):
    spec = _atrp_spec()
    o = make_oracle()
    W.apply(spec, o)
    o.run(50)
    rows = o.atrp_stats()
    assert [r["step"] for r in rows] == [10, 20, 30, 40, 50]
    st = o.get_state("STATE")
    active = int((st == 3).sum())
    assert active == sum(r["activated"] for r in rows) > 0 and all(r["deactivated"] == 0 for r in rows)   # k_deactivate = 0
    dc = 0.2 / 400
    for k, r in enumerate(rows):
        done = sum(q["activated"] for q in rows[:k + 1])
        assert abs(r["ratio_activator"] - max(0.0, 0.2 - dc * done)) < 1e-12
        assert abs(r["ratio_activator"] + r["ratio_deactivator"] - 1.0) < 1e-12        # catalyst is conserved
        assert r["selected"] == 400 and r["candidates"] <= 2 * 3000 // 3 + 200
    # the activator pool runs dry: expected flips per firing fall with ratio_activator
    assert rows[0]["activated"] > rows[-1]["activated"]
    # untouched: particles that match no centre
    assert np.array_equal(st[spec["types"] == 2], spec["state"][spec["types"] == 2])


def test_atrp_activator_equilibrium_with_deactivation(make_oracle):
    spec = _atrp_spec(k_deact=1.0, select_from_all=False, num=300)
    o = make_oracle()
    W.apply(spec, o)
    o.run(400)
    rows = o.atrp_stats()
    assert len(rows) == 40 and sum(r["deactivated"] for r in rows) > 0
    assert all(r["selected"] == 300 for r in rows)               # chosen among the centres only
    st = o.get_state("STATE")
    assert set(np.unique(st[spec["types"] != 2]).tolist()) <= {2, 3}
    assert int((st == 3).sum()) == sum(r["activated"] - r["deactivated"] for r in rows)
    assert all(0.0 <= r["ratio_activator"] <= 1.0 and abs(r["ratio_activator"] + r["ratio_deactivator"] - 1.0) < 1e-12 for r in rows)


def test_atrp_cfg_of_the_shipped_example_parses():
    """examples/atrp_lj/atrp.cfg (golden copy): the [ext_atrp] section reaches the shim's ATRPActivator with the
    four reactive centres of its `options` line."""
    from chemlab_amd.chemlab import reaction_parser
    cfg = reaction_parser.parse_config(os.path.join(GOLD, "atrp_lj", "atrp.cfg"))
    ext = cfg["extensions"]["atrp"]
    assert ext["ext_type"] == "ATRPActivator" and ext["options"].count("->") == 4
    assert cfg["reactions"]["reaction_1"]["extensions"] == ["atrp", "change_neighbour_type"]


def test_mixed_tabulated_follows_the_conversion(tmp_path, oracle_mod):
    """nonbond_params func 10 (gromacs_topology.py:574-583,756-790): U = x * table1 + (1 - x) * table2 with x the chemical
    conversion (particles of a type / total).  Known answer through the shim on the oracle: pair energy and force of two
    particles at a distance between grid points, before and after the conversion observable moves."""
    from chemlab_amd import espp
    r = 0.002 * np.arange(1, 1001)
    t1 = np.stack([r, 3.0 * (1.0 - r / 2.0) ** 2, 3.0 * (1.0 - r / 2.0)], 1)          # U = 3 (1 - r/2)^2, f = -dU/dr
    t2 = np.stack([r, 1.0 * np.exp(-r), 1.0 * np.exp(-r)], 1)
    np.savetxt(tmp_path / "t1.pot", t1, fmt="%15.8g"); np.savetxt(tmp_path / "t2.pot", t2, fmt="%15.8g")
    t1, t2 = np.loadtxt(tmp_path / "t1.pot"), np.loadtxt(tmp_path / "t2.pot")              # (the 8 significant digits of the file format)
    r = t1[:, 0]
    espp.set_engine_factory(lambda: oracle_mod.OracleEngine())
    try:
        system = espp.System()
        system.rng = espp.esutil.RNG(3)
        system.skin = 0.2
        box = (12.0, 12.0, 12.0)
        system.bc = espp.bc.OrthorhombicBC(system.rng, box)
        system.storage = espp.storage.DomainDecomposition(system, espp.tools.decomp.nodeGrid(1), espp.tools.decomp.cellGrid(box, (1, 1, 1), 1.5, 0.2))
        integrator = espp.integrator.VelocityVerlet(system)
        integrator.dt = 1e-5
        d = 0.7313
        plist = [[1, 0, espp.Real3D(3.0, 3.0, 3.0), 1.0], [2, 0, espp.Real3D(3.0 + d, 3.0, 3.0), 1.0]] + \
                [[3 + k, 1, espp.Real3D(8.0, 2.0 + 2.0 * k, 8.0), 1.0] for k in range(4)]          # four far-away particles of type 1
        system.storage.addParticles(plist, "id", "type", "pos", "mass")
        system.storage.decompose()
        vl = espp.VerletList(system, cutoff=1.5, exclusionlist=espp.DynamicExcludeList(integrator, []))
        obs = espp.analysis.ChemicalConversion(system, 2, 4)                                      # fraction of the 4 that became type 2
        mix = espp.interaction.VerletListMixedTabulated(vl)
        mix.setPotential(type1=0, type2=0, potential=espp.interaction.MixedTabulated(1, str(tmp_path / "t1.pot"), str(tmp_path / "t2.pot"), obs, cutoff=1.5))
        system.addInteraction(mix, "lj-mix_tab")
        pe = espp.analysis.PotentialEnergy(system, mix)

        def want(x):
            e = x * np.interp(d, r, t1[:, 1]) + (1 - x) * np.interp(d, r, t2[:, 1])
            f = x * np.interp(d, r, t1[:, 2]) + (1 - x) * np.interp(d, r, t2[:, 2])
            return e, f
        e0, f0 = want(0.0)
        assert abs(pe.compute() - e0) < 1e-9 * abs(e0)
        integrator.run(0)
        assert abs(system.engine.get_state("FORCE")[1, 0] - f0) < 1e-9 * abs(f0)
        system.storage.modifyParticle(4, "type", 2)                                               # conversion 1/4 ...
        system.storage.modifyParticle(5, "type", 2)                                               # ... 2/4
        assert obs.compute() == 0.5                                                               # computing the observable moves the table
        e1, f1 = want(0.5)
        assert abs(pe.compute() - e1) < 1e-9 * abs(e1)
        integrator.run(0)
        assert abs(system.engine.get_state("FORCE")[1, 0] - f1) < 1e-9 * abs(f1)
        # func 12: a constant mixture
        mix.setPotential(type1=0, type2=0, potential=espp.interaction.MixedTabulated(1, table1=str(tmp_path / "t1.pot"), table2=str(tmp_path / "t2.pot"), mix_value=0.8, cutoff=1.5))
        e2, _ = want(0.8)
        assert abs(pe.compute() - e2) < 1e-9 * abs(e2)
    finally:
        from chemlab_amd.engine import Engine
        espp.set_engine_factory(lambda: Engine(device=0, precision=32))


def test_restrict_reaction_only_mapped_pairs_react(make_oracle):
    """RestrictReaction.define_connection (reaction_setup.py:75-78,115-128): with a connectivity map only the listed id
    pairs may react; every event of the restricted reaction is a mapped pair, the other reactions are untouched."""
    spec = W.reactive_melt(n=4000, seed=23, interval=5)
    for r in spec["reaction"]["reactions"]:
        r["rate"] = 1e9
    free, res = make_oracle(), make_oracle()
    W.apply(spec, free); W.apply(spec, res)
    free.run(20)
    ev = free.get_events()
    bonding = ev[ev["reaction"] == 1]
    assert len(bonding) > 20
    allowed = np.stack([bonding["id_a"], bonding["id_b"]], 1)[::2]           # every second bond-forming pair of the free run
    res.reaction_restrict(1, allowed)
    res.run(20)
    ev2 = res.get_events()
    got = {tuple(sorted(p)) for p in np.stack([ev2["id_a"], ev2["id_b"]], 1)[ev2["reaction"] == 1].tolist()}
    assert got and got <= {tuple(sorted(p)) for p in allowed.tolist()}
    assert (ev2["reaction"] == 0).sum() > 0                                  # unrestricted reactions still fire


def _exchange_spec(n_mol=300, seed=31):
    """A-B dimers (bonded, B in state 0) in a bath of free C: the exchange `A(0,1):B(0,1) + C(0,1) -> A(1):C(1) + B(1)` as the
    reference sets it up (reaction_setup.py:167-251): virtual A + C reaction, constraint "A carries a B in state [0,1)",
    B neighbours of A in that window get B's new type / state + 1."""
    rng = np.random.default_rng(seed)
    A, B, C_, B2 = 0, 1, 2, 3
    k = int(np.ceil((3 * n_mol) ** (1.0 / 3.0)))
    a = 1.15
    g = np.arange(k)
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)[:3 * n_mol] * a + 0.5 * a
    sites = sites + rng.uniform(-0.03, 0.03, sites.shape)
    n = 3 * n_mol
    types = np.tile(np.array([A, B, C_], np.int32), n_mol)          # neighbours along the fastest lattice axis: A B C A B C ...
    ids = np.arange(1, n + 1)
    bonds = np.stack([ids[0::3], ids[1::3]], 1)
    return dict(n=n, box=[k * a] * 3, rc=2.5, skin=0.3, dt=0.002, ids=ids, types=types, pos=sites, vel=rng.normal(0, 0.3, (n, 3)),
                mass=np.ones(n), state=np.zeros(n, np.int32), res_id=(np.arange(n) // 3 + 1).astype(np.int32),
                lj=[(i, j, 0.3, 0.9, 2.5) for i in range(4) for j in range(i, 4)], kT=1.0, gamma=1.0, seed=seed,
                lists=[dict(arity=2, kind="HARMONIC", params=[30.0, 1.15], ids=bonds)], exclusions=bonds,
                reaction=dict(interval=5, nearest=True, seed=seed, bond=("HARMONIC", [30.0, 1.0]), type_mass={A: 1.0, B: 1.0, C_: 1.0, B2: 1.0},
                              reactions=[dict(type_1=A, type_2=C_, min_state_1=0, max_state_1=1, min_state_2=0, max_state_2=1, delta_1=1, delta_2=1,
                                              rate=1e9, cutoff=1.6, intramolecular=True, intraresidual=True, is_virtual=True)]))


def _apply_exchange(spec, eng):
    h = W.apply(spec, eng)
    eng.reaction_constraint(0, "type_1", 1, 0, 1)                                   # A must carry a B in state [0, 1)
    eng.reaction_neighbour_change(0, "type_1", 1, 1, 3, 1.0, incr_state=1, state_window=(0, 1))   # that B -> B2, state + 1
    return h


def test_exchange_reaction_constraint_and_incremented_neighbour(make_oracle):
    spec = _exchange_spec()
    o = make_oracle()
    _apply_exchange(spec, o)
    o.run(10)
    ev = o.get_events()
    assert len(ev) > 20
    ty, st = o.get_state("TYPE"), o.get_state("STATE")
    a_ids = ev["id_a"]                                            # role type_1 = A
    assert np.all(spec["types"][a_ids - 1] == 0)
    # every reacted A: state 1, its bonded B became B2 with state 1; unreacted dimers untouched
    assert np.all(st[a_ids - 1] == 1) and np.all(ty[a_ids] == 3) and np.all(st[a_ids] == 1)
    untouched = np.setdiff1d(spec["ids"][0::3], a_ids)
    assert np.all(ty[untouched] == 1) and np.all(st[untouched] == 0)
    assert len(np.unique(a_ids)) == len(a_ids)                    # an A that lost its B (state 1 now) cannot exchange again
    # without the constraint partner (no bonds at all) nothing may react
    spec2 = dict(spec, lists=[], exclusions=None)
    o2 = make_oracle()
    _apply_exchange(spec2, o2)
    o2.run(10)
    assert len(o2.get_events()) == 0
