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
