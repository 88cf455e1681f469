"""Pins the CPU restatement (oracle/) with closed-form answers from doc/topology.rst of the
reference and with F = -grad U by central differences.  The reference ships no numeric
fixtures for this path: these known answers are what anchors the oracle."""
import numpy as np
import pytest

from chemlab_amd.workloads import synthetic_table
from helpers import fd_forces, forces_energy, setup_small, total_epot


def test_lj_two_body_closed_form(make_oracle):
    # doc/topology.rst:12-14  U = 4 eps ((s/r)^12 - (s/r)^6), shifted so that U(rc) = 0
    eps, sig, rc, r = 1.3, 0.9, 2.5, 1.1
    e = setup_small(make_oracle(), [[5, 5, 5], [5 + r, 5, 5]])
    e.nb_lj(0, 0, eps, sig, rc, True)
    f, obs = forces_energy(e)
    sr6 = (sig / r) ** 6
    u = 4 * eps * (sr6 ** 2 - sr6) - 4 * eps * ((sig / rc) ** 12 - (sig / rc) ** 6)
    fmag = 24 * eps * (2 * sr6 ** 2 - sr6) / r
    assert obs["epot_lj"] == pytest.approx(u, rel=1e-13)
    assert f[1, 0] == pytest.approx(fmag, rel=1e-13) and f[0, 0] == pytest.approx(-fmag, rel=1e-13)
    assert np.allclose(f[:, 1:], 0)


def test_lj_cutoff_and_no_interaction_for_sigma_zero(make_oracle):
    e = setup_small(make_oracle(), [[5, 5, 5], [7.6, 5, 5]], skin=0.3)
    e.nb_lj(0, 0, 1.0, 1.0, 2.5, True)
    f, obs = forces_energy(e)
    assert np.all(f == 0) and obs["epot_lj"] == 0
    # sigma <= 0 switches the pair off (gromacs_topology.py:715; chain_growth_catalytic/topol.top:16-17)
    e2 = setup_small(make_oracle(), [[5, 5, 5], [6.0, 5, 5]], types=np.array([0, 1], np.int32))
    e2.nb_lj(0, 0, 1.0, 1.0, 2.5, True)
    e2.nb_lj(0, 1, 0.0, 0.0, 2.5, True)
    f, obs = forces_energy(e2)
    assert np.all(f == 0)


def test_lj_periodic_minimum_image(make_oracle):
    e = setup_small(make_oracle(), [[0.2, 5, 5], [19.7, 5, 5]], box=20.0)
    e.nb_lj(0, 0, 1.0, 1.0, 2.5, False)
    f, obs = forces_energy(e)
    r = 0.5
    assert f[0, 0] == pytest.approx(24 * (2 / r ** 12 - 1 / r ** 6) / r, rel=1e-12)  # pushed to +x
    assert f[0, 0] > 0 and f[1, 0] == pytest.approx(-f[0, 0])


def test_harmonic_bond_closed_form(make_oracle):
    # U = K (r - r0)^2 (ESPResSo++ convention = GROMACS k/2, gromacs_topology.py:918; doc/topology.rst:60)
    K, r0, r = 30.0, 0.97, 1.2
    e = setup_small(make_oracle(), [[5, 5, 5], [5, 5 + r, 5]])
    h = e.list_create(2, "HARMONIC")
    e.list_set_params(h, [K, r0])
    e.list_add(h, [[1, 2]])
    f, obs = forces_energy(e)
    assert obs["epot_list"][0] == pytest.approx(K * (r - r0) ** 2, rel=1e-13)
    assert f[1, 1] == pytest.approx(-2 * K * (r - r0), rel=1e-13)
    assert f[0, 1] == pytest.approx(+2 * K * (r - r0), rel=1e-13)


def test_fene_bond_closed_form(make_oracle):
    K, r0, rmax, r = 30.0, 0.0, 1.5, 1.0
    e = setup_small(make_oracle(), [[5, 5, 5], [5 + r, 5, 5]])
    h = e.list_create(2, "FENE")
    e.list_set_params(h, [K, r0, rmax])
    e.list_add(h, [[1, 2]])
    f, obs = forces_energy(e)
    assert obs["epot_list"][0] == pytest.approx(-0.5 * K * rmax ** 2 * np.log(1 - (r / rmax) ** 2), rel=1e-13)
    assert f[1, 0] == pytest.approx(-K * r / (1 - (r / rmax) ** 2), rel=1e-13)


@pytest.mark.parametrize("kind,params", [("ANG_HARMONIC", [2.5, np.deg2rad(119.0)]), ("ANG_COSINE", [3.0, np.deg2rad(140.0)])])
def test_angle_energy_and_fd(make_oracle, kind, params):
    pos = np.array([[5.9, 5.1, 5.0], [5.0, 5.0, 5.2], [5.3, 5.8, 4.7]])

    def build(e, p):
        setup_small(e, p)
        h = e.list_create(3, kind)
        e.list_set_params(h, params)
        e.list_add(h, [[1, 2, 3]])
    e = make_oracle()
    build(e, pos)
    f, obs = forces_energy(e)
    r1, r2 = pos[0] - pos[1], pos[2] - pos[1]
    th = np.arccos(r1 @ r2 / np.linalg.norm(r1) / np.linalg.norm(r2))
    u = params[0] * (th - params[1]) ** 2 if kind == "ANG_HARMONIC" else params[0] * (1 + np.cos(th - params[1]))
    assert obs["epot_list"][0] == pytest.approx(u, rel=1e-12)
    assert np.allclose(f, fd_forces(make_oracle, build, pos), rtol=1e-6, atol=1e-7)
    assert np.allclose(f.sum(0), 0, atol=1e-12)


@pytest.mark.parametrize("kind,params", [("DIH_NCOS", [2.0, np.deg2rad(30.0), 3.0]), ("DIH_RB", [1.0, -0.5, 0.8, 0.3, -0.2, 0.1])])
def test_dihedral_energy_and_fd(make_oracle, kind, params):
    pos = np.array([[5.0, 5.0, 5.0], [5.9, 5.2, 5.1], [6.2, 6.1, 5.4], [7.1, 6.3, 6.2]])

    def build(e, p):
        setup_small(e, p)
        h = e.list_create(4, kind)
        e.list_set_params(h, params)
        e.list_add(h, [[1, 2, 3, 4]])
    e = make_oracle()
    build(e, pos)
    f, obs = forces_energy(e)
    b1, b2, b3 = pos[1] - pos[0], pos[2] - pos[1], pos[3] - pos[2]
    m, n = np.cross(b1, b2), np.cross(b2, b3)
    phi = np.arctan2(np.linalg.norm(b2) * (b1 @ n), m @ n)
    if kind == "DIH_NCOS":
        u = params[0] * (1 + np.cos(params[2] * phi - params[1]))
    else:
        u = sum(c * np.cos(phi - np.pi) ** k for k, c in enumerate(params))
    assert obs["epot_list"][0] == pytest.approx(u, rel=1e-12)
    assert np.allclose(f, fd_forces(make_oracle, build, pos), rtol=1e-6, atol=1e-7)
    assert np.allclose(f.sum(0), 0, atol=1e-12)


def test_table_linear_interpolation(make_oracle):
    # Tabulated(itype=1): linear between rows of the r e f table (.pot rows start at dr)
    r0, dr, etab, ftab = synthetic_table(nrow=750, dr=0.002, rc=1.4)
    r = 0.7313
    e = setup_small(make_oracle(), [[5, 5, 5], [5, 5, 5 + r]], rc=1.4, skin=0.1)
    e.nb_table(0, 0, r0, dr, etab, ftab, 1.4)
    f, obs = forces_energy(e)
    t = (r - r0) / dr
    k = int(t)
    w = t - k
    assert obs["epot_tab"] == pytest.approx(etab[k] + w * (etab[k + 1] - etab[k]), rel=1e-12)
    assert f[1, 2] == pytest.approx(ftab[k] + w * (ftab[k + 1] - ftab[k]), rel=1e-12)
    assert f[0, 2] == pytest.approx(-f[1, 2])


def test_typed_list_uses_current_types(make_oracle):
    # "Types" lists pick parameters from the particle types at evaluation time (gromacs_topology.py:969-981)
    e = setup_small(make_oracle(), [[5, 5, 5], [6.1, 5, 5], [7.3, 5, 5]], types=np.array([0, 1, 1], np.int32))
    h = e.list_create(2, "HARMONIC", by_types=True)
    e.list_set_params(h, [10.0, 1.0], types=(0, 1))
    e.list_set_params(h, [20.0, 1.1], types=(1, 1))
    e.list_add(h, [[1, 2], [2, 3]])
    _, obs = forces_energy(e)
    assert obs["epot_list"][0] == pytest.approx(10.0 * 0.1 ** 2 + 20.0 * 0.1 ** 2, rel=1e-10)


def test_cap_force_rescales_conservative_force_only(make_oracle):
    """integrator.CapForce (start_simulation.py:320-324): |f| capped at max_force, direction kept; the cap is
    applied before the thermostat, so with gamma > 0 the friction term comes on top of the capped force."""
    import numpy as np
    spec = dict(n=2, box=[20.0, 20.0, 20.0], rc=2.5, skin=0.3, dt=0.001, ids=np.array([1, 2]), types=np.zeros(2, np.int32),
                pos=np.array([[5.0, 5.0, 5.0], [5.9, 5.0, 5.0]]), vel=np.array([[1.0, 0, 0], [0, 0, 0]]), mass=np.ones(2),
                lj=[(0, 0, 1.0, 1.0, 2.5)], kT=1.0, gamma=0.0, seed=1)
    from chemlab_amd import workloads as W
    o = make_oracle(); W.apply(spec, o, thermostat=False)
    o.run(0)
    f0 = o.get_state("FORCE")
    mag = np.linalg.norm(f0[0])
    assert mag > 50                                          # r = 0.9 sigma: strongly repulsive
    o2 = make_oracle(); W.apply(spec, o2, thermostat=False); o2.cap_force(10.0)
    o2.run(0)
    f1 = o2.get_state("FORCE")
    assert np.allclose(np.linalg.norm(f1, axis=1), 10.0)
    assert np.allclose(f1 / 10.0, f0 / mag)
    # one step: dv = dt * f_capped / m (both half kicks use capped forces of similar size)
    o2.run(1)
    v = o2.get_state("VEL")
    assert abs((v[0, 0] - 1.0) + 0.001 * 10.0) < 1e-3 * 0.001 * 10.0 + 1e-9


def test_tabulated_bond_reproduces_the_function_it_samples(make_oracle):
    """Tabulated(itype=1) on a FixedPairList (bond func 8, gromacs_topology.py:919-925): a table sampled from
    U = K (r - r0)^2, f = -dU/dr must give the harmonic bond's force up to the linear-interpolation error,
    exactly at the grid points, and clamp to its end rows outside the grid."""
    import numpy as np
    from chemlab_amd import workloads as W
    K, r0, dr = 30.0, 0.97, 0.002
    r = dr * np.arange(1, 1001)
    e, f = K * (r - r0) ** 2, -2.0 * K * (r - r0)
    base = dict(n=2, box=[20.0, 20.0, 20.0], rc=2.5, skin=0.3, dt=0.001, ids=np.array([1, 2]), types=np.zeros(2, np.int32),
                vel=np.zeros((2, 3)), mass=np.ones(2), kT=1.0, gamma=0.0, seed=1, exclusions=np.array([[1, 2]]))
    for sep, tol in ((1.1, 0.0), (1.1011, 1e-4), (0.5003, 1e-4)):       # grid point, between grid points (twice)
        spec = dict(base, pos=np.array([[5.0, 5.0, 5.0], [5.0 + sep, 5.0, 5.0]]))
        a, b = make_oracle(), make_oracle()
        W.apply(spec, a, thermostat=False); W.apply(spec, b, thermostat=False)
        ha = a.list_create(2, "HARMONIC"); a.list_set_params(ha, [K, r0]); a.list_add(ha, [[1, 2]])
        hb = b.list_create(2, "TABULATED"); b.list_set_params(hb, [b.table_create(r[0], dr, e, f)]); b.list_add(hb, [[1, 2]])
        a.run(0); b.run(0)
        fa, fb = a.get_state("FORCE"), b.get_state("FORCE")
        assert np.abs(fa - fb).max() <= tol * np.abs(fa).max() + 1e-12
        assert np.allclose(fb[0], -fb[1])
        assert b.observe()["epot_list"][0] == pytest.approx(a.observe()["epot_list"][0], rel=max(tol, 1e-12) * 10, abs=1e-6)
    # beyond the last row (r = 2.0 + ...): end row, f(r_last)/r along the bond
    spec = dict(base, pos=np.array([[5.0, 5.0, 5.0], [7.5, 5.0, 5.0]]))
    b = make_oracle(); W.apply(spec, b, thermostat=False)
    hb = b.list_create(2, "TABULATED"); b.list_set_params(hb, [b.table_create(r[0], dr, e, f)]); b.list_add(hb, [[1, 2]])
    b.run(0)
    assert b.get_state("FORCE")[0, 0] == pytest.approx(-f[-1], rel=1e-12)


def test_tabulated_angle_reproduces_the_function_it_samples(make_oracle, tmp_path):
    """TabulatedAngular(itype=1) (angle func 8, gromacs_topology.py:1074-1080): a table of U = K (theta - theta0)^2,
    -dU/dtheta in radians gives the harmonic angle's forces; and the .xvg (degrees) -> .pot (radians) conversion
    of tools/convert_gromacs2espp.py:73-83 (x -> radians, f -> f*180/pi, rows 0 < theta <= pi)."""
    import numpy as np
    from chemlab_amd import workloads as W
    from chemlab_amd.chemlab import tables
    K, th0 = 40.0, np.deg2rad(110.0)
    deg = np.arange(0, 181)                                          # GROMACS angle table: one row per degree, 0..180
    xvg = tmp_path / "table_a3.xvg"
    with open(xvg, "w") as fh:
        for d in deg:
            th = np.deg2rad(d)
            fh.write("%g %.12g %.12g\n" % (d, K * (th - th0) ** 2, -2.0 * K * (th - th0) * np.pi / 180.0))   # f per degree
    pot = tmp_path / "table_a3.pot"
    assert tables.convert_table(str(xvg), str(pot)) == 180           # theta = 0 dropped
    tab = np.loadtxt(pot)
    assert tab[0, 0] == pytest.approx(np.deg2rad(1.0)) and tab[-1, 0] == pytest.approx(np.pi)
    assert np.allclose(tab[:, 2], -2.0 * K * (tab[:, 0] - th0), rtol=1e-6, atol=1e-5)   # %15.8g: 8 significant digits
    pos = np.array([[5.0, 5.0, 5.0], [6.0, 5.0, 5.0], [6.0 + np.cos(np.deg2rad(97.3)) * -1.0, 5.0 + np.sin(np.deg2rad(97.3)), 5.0]])
    base = dict(n=3, box=[20.0, 20.0, 20.0], rc=2.5, skin=0.3, dt=0.001, ids=np.array([1, 2, 3]), types=np.zeros(3, np.int32), pos=pos,
                vel=np.zeros((3, 3)), mass=np.ones(3), kT=1.0, gamma=0.0, seed=1, exclusions=np.array([[1, 2], [2, 3], [1, 3]]))
    a, b = make_oracle(), make_oracle()
    W.apply(base, a, thermostat=False); W.apply(base, b, thermostat=False)
    ha = a.list_create(3, "ANG_HARMONIC"); a.list_set_params(ha, [K, th0]); a.list_add(ha, [[1, 2, 3]])
    hb = b.list_create(3, "ANG_TABULATED"); b.list_set_params(hb, [b.table_create(tab[0, 0], tab[1, 0] - tab[0, 0], tab[:, 1], tab[:, 2])]); b.list_add(hb, [[1, 2, 3]])
    a.run(0); b.run(0)
    fa, fb = a.get_state("FORCE"), b.get_state("FORCE")
    assert np.abs(fa).max() > 1.0
    assert np.abs(fa - fb).max() < 2e-3 * np.abs(fa).max()          # linear interpolation on a one-degree grid
    assert np.abs(fb.sum(0)).max() < 1e-12


def test_tabulated_dihedral_reproduces_the_function_it_samples(make_oracle):
    """TabulatedDihedral(itype=1) (dihedral func 8, gromacs_topology.py:1192-1198): a table of U = K (1 + cos(n phi - phi0))
    and -dU/dphi over [-pi, pi] gives the DihedralHarmonicNCos forces up to the interpolation error."""
    import numpy as np
    from chemlab_amd import workloads as W
    K, phi0, mult = 1.5, np.deg2rad(20.0), 3.0
    dphi = 2 * np.pi / 1440
    phi = -np.pi + dphi * np.arange(1441)
    e, f = K * (1.0 + np.cos(mult * phi - phi0)), K * mult * np.sin(mult * phi - phi0)
    pos = np.array([[5.0, 5.0, 5.0], [5.9, 5.2, 5.1], [6.2, 6.1, 5.4], [7.1, 6.3, 6.2]])
    base = dict(n=4, box=[20.0, 20.0, 20.0], rc=2.5, skin=0.3, dt=0.001, ids=np.arange(1, 5), types=np.zeros(4, np.int32), pos=pos,
                vel=np.zeros((4, 3)), mass=np.ones(4), kT=1.0, gamma=0.0, seed=1,
                exclusions=np.array([[1, 2], [2, 3], [3, 4], [1, 3], [2, 4], [1, 4]]))
    a, b = make_oracle(), make_oracle()
    W.apply(base, a, thermostat=False); W.apply(base, b, thermostat=False)
    ha = a.list_create(4, "DIH_NCOS"); a.list_set_params(ha, [K, phi0, mult]); a.list_add(ha, [[1, 2, 3, 4]])
    hb = b.list_create(4, "DIH_TABULATED"); b.list_set_params(hb, [b.table_create(phi[0], dphi, e, f)]); b.list_add(hb, [[1, 2, 3, 4]])
    a.run(0); b.run(0)
    fa, fb = a.get_state("FORCE"), b.get_state("FORCE")
    assert np.abs(fa).max() > 0.1
    assert np.abs(fa - fb).max() < 1e-4 * np.abs(fa).max()
    assert b.observe()["epot_list"][0] == pytest.approx(a.observe()["epot_list"][0], rel=1e-4)


def test_berendsen_and_isokinetic_rescaling(make_oracle):
    """integrator.BerendsenThermostat / Isokinetic (start_simulation.py:341-348): after a step the kinetic temperature
    kT = 2 Ekin / (3 N) is pulled towards the target by v *= sqrt(1 + dt/tau (kT0/kT - 1)), resp. set to it exactly."""
    import numpy as np
    from chemlab_amd import workloads as W
    spec = W.lj_melt(n=500, seed=4, kT=1.0)
    kT0 = 0.6
    o = make_oracle(); W.apply(spec, o, thermostat=False)
    o.thermostat_rescale("isokinetic", kT0, 5)
    o.run(5)
    assert o.observe()["temperature"] == pytest.approx(kT0, rel=1e-12)       # exactly on target right after a coupling step
    o.run(3)
    assert abs(o.observe()["temperature"] - kT0) > 1e-6                      # free flight in between
    a, b = make_oracle(), make_oracle()
    W.apply(spec, a, thermostat=False); W.apply(spec, b, thermostat=False)
    tau = 0.05
    b.thermostat_rescale("berendsen", kT0, tau)
    a.run(1); b.run(1)
    kTa, va, vb = a.observe()["temperature"], a.get_state("VEL"), b.get_state("VEL")
    lam = np.sqrt(1.0 + spec["dt"] / tau * (kT0 / kTa - 1.0))
    assert np.allclose(vb, lam * va, rtol=1e-12, atol=1e-14)                 # one step: same trajectory, scaled at the end
    b.run(400)
    assert b.observe()["temperature"] == pytest.approx(kT0, rel=0.05)        # relaxed to the target


def test_stochastic_velocity_rescaling_samples_canonical_kinetic_energy(make_oracle):
    """integrator.StochasticVelocityRescaling (start_simulation.py:337-340; Bussi-Donadio-Parrinello 2007): the kinetic
    energy is pulled to the target and FLUCTUATES canonically: <kT> = kT0, var(K)/<K>^2 = 2 / (3N)."""
    import numpy as np
    from chemlab_amd import workloads as W
    spec = W.lj_melt(n=500, seed=4, kT=1.0)
    kT0, n = 0.6, 500
    o = make_oracle(); W.apply(spec, o, thermostat=False)
    o.thermostat_svr(kT0, 0.05, 7)
    o.run(300)
    ts = []
    for _ in range(400):
        o.run(5)
        ts.append(o.observe()["temperature"])
    ts = np.array(ts)
    assert ts.mean() == pytest.approx(kT0, rel=0.01)
    assert ts.std() / ts.mean() == pytest.approx(np.sqrt(2.0 / (3 * n)), rel=0.25)
    # keyed stream: the same seed reproduces the trajectory, another seed does not
    a, b, c = make_oracle(), make_oracle(), make_oracle()
    for e, seed in ((a, 11), (b, 11), (c, 12)):
        W.apply(spec, e, thermostat=False); e.thermostat_svr(kT0, 0.05, seed); e.run(20)
    assert np.array_equal(a.get_state("VEL"), b.get_state("VEL"))
    assert not np.allclose(a.get_state("VEL"), c.get_state("VEL"))
    # coupling <= 0 switches it off: plain NVE afterwards
    a.thermostat_svr(kT0, 0.0, 11)
    d = make_oracle(); W.apply(spec, d, thermostat=False)
    d.set_particles(spec["ids"], spec["types"], a.get_state("POS"), spec["mass"], vel=a.get_state("VEL"))
    a.run(0); d.run(0)
    assert np.allclose(a.get_state("FORCE"), d.get_state("FORCE"), rtol=1e-9, atol=1e-9)


def test_svr_single_draw_matches_the_formula():
    """One SVR factor recomputed in numpy from the same Philox stream (include/chem_philox.h svr_lambda): with
    taut -> infinity the factor is 1, with taut -> 0 the new kinetic energy is K_ref chi^2(ndeg)/ndeg."""
    import ctypes as C
    from oracle import oracle
    lib = C.CDLL(oracle.build())
    lib.orc_svr_lambda.restype = C.c_double
    lib.orc_svr_lambda.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_int64, C.c_double]
    assert lib.orc_svr_lambda(1, 2, 3.0, 5.0, 300, 1e12) == pytest.approx(1.0, abs=1e-5)
    lams = np.array([lib.orc_svr_lambda(9, s, 2.0, 2.0, 3000, 1e-3) for s in range(2000)])
    k = 2.0 * lams ** 2          # K_new with instantaneous coupling: K_ref * chi^2(ndeg) / ndeg
    assert k.mean() == pytest.approx(2.0, rel=2e-3)
    assert k.std() / k.mean() == pytest.approx(np.sqrt(2.0 / 3000), rel=0.06)


def test_lj_14_pair_list_closed_form_and_shift(make_oracle):
    """FixedPairListLennardJones (1-4 pairs, gromacs_topology.py:1314-1411): the pair feels LJ through the list although it is
    excluded from the non-bonded list; energy shifted to zero at the cutoff (espressopp's LennardJones default shift='auto')."""
    eps, sig, rc, r = 0.7, 0.95, 2.0, 1.25
    e = setup_small(make_oracle(), [[5, 5, 5], [5 + r, 5, 5]])
    e.nb_lj(0, 0, 1.0, 1.0, 2.5, True)
    e.set_exclusions([[1, 2]])
    h = e.list_create(2, "LJ_BOND")
    e.list_set_params(h, [eps, sig, rc])
    e.list_add(h, [[1, 2]])
    f, obs = forces_energy(e)
    sr6, sc6 = (sig / r) ** 6, (sig / rc) ** 6
    assert obs["epot_lj"] == 0.0
    assert obs["epot_list"][h] == pytest.approx(4 * eps * ((sr6 ** 2 - sr6) - (sc6 ** 2 - sc6)), rel=1e-13)
    assert f[1, 0] == pytest.approx(24 * eps * (2 * sr6 ** 2 - sr6) / r, rel=1e-13)
    # beyond the list potential's own cutoff: nothing
    e2 = setup_small(make_oracle(), [[5, 5, 5], [5 + 2.2, 5, 5]])
    e2.nb_lj(0, 0, 1.0, 1.0, 2.5, True); e2.set_exclusions([[1, 2]])
    h2 = e2.list_create(2, "LJ_BOND"); e2.list_set_params(h2, [eps, sig, rc]); e2.list_add(h2, [[1, 2]])
    f2, obs2 = forces_energy(e2)
    assert np.all(f2 == 0) and obs2["epot_list"][h2] == 0.0
