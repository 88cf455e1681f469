"""Synthetic inputs of the BASELINE configurations.

Every generator returns a plain `spec` dict (numpy arrays + scalars); `apply(spec, engine)`
issues the C-ABI calls on an Engine.  LJ reduced units (eps=sigma=m=kB=1) unless stated.
Seeds are fixed so CPU oracle and HIP path see identical inputs.

  lj_melt        C2  32k LJ melt, rho*=0.8, rc=2.5         (examples/*_lj params shape)
  polymer_melt   C3  bead-spring chains, tabulated non-bonded + harmonic bonds/angles
                     (examples/mf/espp_cg_1 shape)
  reactive_melt  C4/C5 chain-growth reactive LJ (examples/chain_growth_catalytic shape:
                     topol.top:4-25, reaction.cfg, params)
  trimer_melt    examples/atrp_lj shape (MA-ML-MA trimers, bonds+angles+exclusions)
"""
import numpy as np


def _lattice(n_target, rho):
    """fcc (N=4k^3) or simple cubic (N=k^3) lattice at number density rho. Returns pos, L, kind."""
    k = round((n_target / 4.0) ** (1.0 / 3.0))
    if 4 * k ** 3 == n_target:
        L = (n_target / rho) ** (1.0 / 3.0)
        a = L / k
        g = np.arange(k)
        cx, cy, cz = np.meshgrid(g, g, g, indexing="ij")
        cell = np.stack([cx.ravel(), cy.ravel(), cz.ravel()], 1).astype(np.float64)
        basis = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]])
        pos = (cell[:, None, :] + basis[None, :, :]).reshape(-1, 3) * a
        return pos + 0.25 * a, L, "fcc"
    k = round(n_target ** (1.0 / 3.0))
    if k ** 3 != n_target:
        raise ValueError("N must be 4k^3 (fcc) or k^3 (sc), got %d" % n_target)
    L = (n_target / rho) ** (1.0 / 3.0)
    a = L / k
    g = np.arange(k)
    cx, cy, cz = np.meshgrid(g, g, g, indexing="ij")
    pos = np.stack([cx.ravel(), cy.ravel(), cz.ravel()], 1).astype(np.float64) * a + 0.5 * a
    return pos, L, "sc"


def _maxwell(rng, n, kT, mass):
    v = rng.standard_normal((n, 3)) * np.sqrt(kT / np.asarray(mass, dtype=np.float64))[:, None]
    m = np.asarray(mass, dtype=np.float64)[:, None]
    v -= (m * v).sum(0) / m.sum()
    return v


def lj_melt(n=32000, rho=0.8, rc=2.5, skin=0.3, dt=0.005, kT=1.0, seed=1, gamma=0.0, jitter=0.02):
    """C2: single-type LJ melt on an fcc lattice with Maxwell velocities, zero COM momentum."""
    rng = np.random.default_rng(seed)
    pos, L, _ = _lattice(n, rho)
    pos = pos + rng.uniform(-jitter, jitter, pos.shape)
    mass = np.ones(n)
    return dict(name="lj_melt", n=n, box=[L] * 3, rc=rc, skin=skin, dt=dt,
                ids=np.arange(1, n + 1), types=np.zeros(n, np.int32), pos=pos, vel=_maxwell(rng, n, kT, mass),
                mass=mass, state=np.zeros(n, np.int32), res_id=np.arange(1, n + 1, dtype=np.int32),
                lj=[(0, 0, 1.0, 1.0, rc)], kT=kT, gamma=gamma, seed=seed)


def reactive_melt(n=256000, rho=0.8442, rc=2.5, skin=0.3, dt=0.005, kT=0.5, gamma=5.0, seed=2,
                  interval=500, rate=1.0, rcut_react=1.2, jitter=0.05):
    """C4/C5: n/2 'MOL' molecules = unbonded (A,B) pairs (chain_growth_catalytic/topol.top:22-25),
    initial state 1 (topol.top:4-7), LJ eps=sigma=1 for all pairs except A-B and A-D
    (topol.top:15-17), the four reactions of reaction.cfg, new bonds Harmonic K=30 r0=0.97."""
    rng = np.random.default_rng(seed)
    pos, L, kind = _lattice(n, rho)
    pos = pos + rng.uniform(-jitter, jitter, pos.shape)
    A, B, D = 0, 1, 2
    types = np.empty(n, np.int32)
    if kind == "fcc":
        types[:] = np.tile(np.array([A, B, A, B], np.int32), n // 4)
    else:
        k = round(n ** (1.0 / 3.0))
        g = np.arange(k)
        cx, cy, cz = np.meshgrid(g, g, g, indexing="ij")
        # molecule = two z-adjacent sites (z is the fastest index); A sits on the even-parity site
        par = ((cx + cy + cz) % 2).ravel()
        types[:] = np.where(par == 0, A, B)
        if k % 2:
            raise ValueError("sc reactive melt needs an even lattice")
    res_id = (np.arange(n) // 2 + 1).astype(np.int32)
    mass = np.ones(n)
    lj = [(A, A, 1.0, 1.0, rc), (B, B, 1.0, 1.0, rc), (D, D, 1.0, 1.0, rc), (B, D, 1.0, 1.0, rc)]
    reactions = [  # reaction.cfg: a, b, c, d
        dict(type_1=A, type_2=B, min_state_1=1, max_state_1=2, min_state_2=1, max_state_2=2, delta_1=2, delta_2=0,
             new_type_2=D, is_virtual=True),
        dict(type_1=A, type_2=A, min_state_1=3, max_state_1=4, min_state_2=1, max_state_2=2, delta_1=1, delta_2=1),
        dict(type_1=A, type_2=A, min_state_1=2, max_state_1=3, min_state_2=1, max_state_2=2, delta_1=1, delta_2=1),
        dict(type_1=A, type_2=D, min_state_1=3, max_state_1=4, min_state_2=1, max_state_2=2, delta_1=-2, delta_2=0,
             new_type_2=B, is_virtual=True),
    ]
    for r in reactions:
        r.update(rate=rate, cutoff=rcut_react, intramolecular=False, intraresidual=False)
        r.setdefault("is_virtual", False)
    return dict(name="reactive_melt", n=n, box=[L] * 3, rc=rc, skin=skin, dt=dt,
                ids=np.arange(1, n + 1), types=types, pos=pos, vel=_maxwell(rng, n, kT, mass), mass=mass,
                state=np.ones(n, np.int32), res_id=res_id, lj=lj, kT=kT, gamma=gamma, seed=seed,
                reaction=dict(interval=interval, nearest=True, seed=seed, bond=("HARMONIC", [30.0, 0.97]),
                              reactions=reactions, type_mass={A: 1.0, B: 1.0, D: 1.0}))


def synthetic_table(nrow=1750, dr=0.002, rc=1.5, eps=2.0, sigma=0.6):
    """Smooth, purely analytic stand-in for a GROMACS-converted CG table (r e f rows of a .pot
    file; tools/convert_gromacs2espp.py:84-107 drops r=0, so rows start at dr).  Soft-core
    repulsion + shallow well, force = -dU/dr evaluated analytically."""
    r = dr * np.arange(1, nrow + 1)
    x = r / sigma
    e = eps * (np.exp(-2.0 * (x - 1.0) * 3.0) - 2.0 * np.exp(-(x - 1.0) * 3.0))
    f = eps * (3.0 / sigma) * (2.0 * np.exp(-2.0 * (x - 1.0) * 3.0) - 2.0 * np.exp(-(x - 1.0) * 3.0))
    sw = np.where(r < rc, (1.0 - (r / rc) ** 2) ** 2, 0.0)  # smooth cut
    dsw = np.where(r < rc, -4.0 * r / rc ** 2 * (1.0 - (r / rc) ** 2), 0.0)
    return dr, dr, e * sw, f * sw - e * dsw


def polymer_melt(n_chains=4000, chain_len=32, rho=3.59, rc=1.5, skin=0.1, dt=0.002, seed=3,
                 kT=800 * 0.0083144621, gamma=10.0, mass=216.2, r0=0.7, K=159828.8,
                 theta0_deg=119.0, Ka=244.0):
    """C3: bead-spring chains laid along a boustrophedon path through a simple-cubic lattice,
    harmonic bonds/angles (mf/espp_cg_1/topol.top bondtypes), one tabulated non-bonded pair."""
    n = n_chains * chain_len
    k = int(np.ceil(n ** (1.0 / 3.0)))
    if k % 2:
        k += 1
    L = (n / rho) ** (1.0 / 3.0)
    a = L / k
    rng = np.random.default_rng(seed)
    # snake through the lattice so consecutive sites are neighbours
    sites = []
    for z in range(k):
        ys = range(k) if z % 2 == 0 else range(k - 1, -1, -1)
        for yi, y in enumerate(ys):
            fwd = (yi + z * k) % 2 == 0
            xs = range(k) if fwd else range(k - 1, -1, -1)
            for x in xs:
                sites.append((x, y, z))
    sites = np.array(sites[:n], dtype=np.float64)
    pos = sites * a + 0.5 * a + rng.uniform(-0.05 * a, 0.05 * a, (n, 3))
    ids = np.arange(1, n + 1)
    first = np.arange(n).reshape(n_chains, chain_len)
    bonds = np.stack([first[:, :-1].ravel(), first[:, 1:].ravel()], 1) + 1
    angles = np.stack([first[:, :-2].ravel(), first[:, 1:-1].ravel(), first[:, 2:].ravel()], 1) + 1
    excl = [bonds, angles[:, [0, 2]]]
    quad = np.stack([first[:, :-3].ravel(), first[:, 3:].ravel()], 1) + 1  # nrexcl 3
    excl.append(quad)
    m = np.full(n, mass)
    r0t, drt, e, f = synthetic_table(rc=rc)
    return dict(name="polymer_melt", n=n, box=[L] * 3, rc=rc, skin=skin, dt=dt, ids=ids,
                types=np.zeros(n, np.int32), pos=pos, vel=_maxwell(rng, n, kT, m), mass=m,
                state=np.zeros(n, np.int32), res_id=(np.arange(n) // chain_len + 1).astype(np.int32),
                tables=[(0, 0, r0t, drt, e, f, rc)], kT=kT, gamma=gamma, seed=seed,
                lists=[dict(arity=2, kind="HARMONIC", params=[K, r0], ids=bonds),
                       dict(arity=3, kind="ANG_HARMONIC", params=[Ka, np.deg2rad(theta0_deg)], ids=angles)],
                exclusions=np.concatenate(excl, 0))


def trimer_melt(n_mol=200, rho=0.27, rc=2.5, skin=0.4, dt=0.0025, kT=1.0, gamma=1.0, seed=4,
                interval=200, reactive=True):
    """examples/atrp_lj shape: MA-ML-MA trimers (topol.top:38), harmonic bonds 0.97/K=30 and
    angles 180deg/K=1.25 (ffnb.itp:12,24), exclusions nrexcl 2, LJ eps=sigma=1.
    With reactive=True: end-group coupling MA(0,1)+MA(0,1) -> MA(1):MA(1) forming bonds, and
    the topology manager spawns ML-MA-MA / MA-MA-ML angles for registered type triples."""
    rng = np.random.default_rng(seed)
    n = 3 * n_mol
    # trimers lie along x on an orthorhombic lattice so that no two beads overlap at t=0:
    # cell (3.2, 1.86, 1.86) * s leaves a 1.26 end-to-end gap along x (reactive MA ends face each other)
    k = int(np.ceil(n_mol ** (1.0 / 3.0)))
    cell = np.array([3.2, 1.86, 1.86])
    cell *= ((3.0 / rho) / cell.prod()) ** (1.0 / 3.0)
    box = (k * cell).tolist()
    g = np.arange(k)
    cx, cy, cz = np.meshgrid(g, g, g, indexing="ij")
    centres = (np.stack([cx.ravel(), cy.ravel(), cz.ravel()], 1)[:n_mol] + 0.5) * cell
    dirs = np.array([1.0, 0.0, 0.0]) + rng.uniform(-0.05, 0.05, (n_mol, 3))
    dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    pos = np.empty((n_mol, 3, 3))
    pos[:, 1] = centres
    pos[:, 0] = centres - 0.97 * dirs
    pos[:, 2] = centres + 0.97 * dirs
    pos = pos.reshape(n, 3) + rng.uniform(-0.02, 0.02, (n, 3))
    MA, ML = 0, 1
    types = np.tile(np.array([MA, ML, MA], np.int32), n_mol)
    ids = np.arange(1, n + 1)
    b0 = np.arange(n_mol) * 3 + 1
    bonds = np.concatenate([np.stack([b0, b0 + 1], 1), np.stack([b0 + 1, b0 + 2], 1)])
    angles = np.stack([b0, b0 + 1, b0 + 2], 1)
    excl = np.concatenate([bonds, np.stack([b0, b0 + 2], 1)])
    mass = np.ones(n)
    spec = dict(name="trimer_melt", n=n, box=box, rc=rc, skin=skin, dt=dt, ids=ids, types=types, pos=pos,
                vel=_maxwell(rng, n, kT, mass), mass=mass, state=np.zeros(n, np.int32),
                res_id=(np.arange(n) // 3 + 1).astype(np.int32),
                lj=[(MA, MA, 1.0, 1.0, rc), (MA, ML, 1.0, 1.0, rc), (ML, ML, 1.0, 1.0, rc)],
                kT=kT, gamma=gamma, seed=seed,
                lists=[dict(arity=2, kind="HARMONIC", params=[30.0, 0.97], ids=bonds),
                       dict(arity=3, kind="ANG_HARMONIC", params=[1.25, np.pi], ids=angles,
                            register=[(ML, MA, MA)])],
                exclusions=excl)
    if reactive:
        spec["reaction"] = dict(
            interval=interval, nearest=True, seed=seed, bond=("HARMONIC", [30.0, 0.97]),
            reactions=[dict(type_1=MA, type_2=MA, min_state_1=0, max_state_1=1, min_state_2=0, max_state_2=1,
                            delta_1=1, delta_2=1, rate=1.0e9, cutoff=1.2, intramolecular=False,
                            intraresidual=False, is_virtual=False)],
            type_mass={MA: 1.0, ML: 1.0})
    return spec


def snap_to_grid(spec):
    """Positions exactly representable in BOTH precisions of the engine: the fp32 build stores int32 fixed-point coordinates
    q = rint((x - L/2) / s), s = L / 2^31 (md_kernels.hpp "position codec") and decodes them as q * s + L/2 in fp64, which
    is what this returns -- frozen-configuration tests feed the same bits to the oracle and to both builds."""
    L = np.asarray(spec["box"], dtype=np.float64)
    s = L / 2147483648.0
    x = np.asarray(spec["pos"], dtype=np.float64)
    xf = x - np.floor(x / L) * L
    q = np.rint((xf - 0.5 * L) / s)
    q = np.where(q >= 2 ** 30, q - 2 ** 31, q)
    out = dict(spec)
    out["pos"] = q * s + 0.5 * L
    return out


def apply(spec, eng, thermostat=True, reactions=True):
    """Issue the set-up calls for `spec` on Engine `eng`; returns dict of list handles."""
    eng.set_box(spec["box"])
    if "rebuild_criterion" in spec:
        eng.set_option("rebuild_criterion", spec["rebuild_criterion"])
    eng.set_cutoff(spec["rc"], spec["skin"])
    eng.set_dt(spec["dt"])
    eng.set_particles(spec["ids"], spec["types"], spec["pos"], spec["mass"], vel=spec.get("vel"),
                      state=spec.get("state"), res_id=spec.get("res_id"))
    for (t1, t2, eps, sig, rc) in spec.get("lj", []):
        eng.nb_lj(t1, t2, eps, sig, rc, True)
    for (t1, t2, r0, dr, e, f, rc) in spec.get("tables", []):
        eng.nb_table(t1, t2, r0, dr, e, f, rc)
    handles = {}
    for i, l in enumerate(spec.get("lists", [])):
        h = eng.list_create(l["arity"], l["kind"], False)
        eng.list_set_params(h, l["params"])
        eng.list_add(h, l["ids"])
        for reg in l.get("register", []):
            eng.topology_register(h, reg)
        handles[i] = h
    if spec.get("exclusions") is not None:
        eng.set_exclusions(spec["exclusions"])
    if thermostat and spec.get("gamma", 0) > 0:
        eng.thermostat_langevin(spec["kT"], spec["gamma"], spec["seed"])
        if spec.get("thermal_types") is not None:       # thermal groups (start_simulation.py:312-336)
            eng.thermostat_langevin_types(spec["thermal_types"])
    rx = spec.get("reaction")
    if rx and reactions:
        hb = eng.list_create(2, rx["bond"][0], False)
        eng.list_set_params(hb, rx["bond"][1])
        handles["reaction_bonds"] = hb
        eng.reaction_init(rx["interval"], rx["nearest"], rx.get("max_per_interval", 0), rx["seed"])
        for r in rx["reactions"]:
            r = dict(r)
            for k in (1, 2):
                nt = r.get("new_type_%d" % k, -1)
                if nt >= 0:
                    r["new_mass_%d" % k] = rx["type_mass"][nt]
            eng.reaction_add(bond_list=hb, **r)
        eng.reactions_enable(True)
    at = spec.get("atrp")
    if at and reactions:        # ATRPActivator (reaction_post_process.py:380-426), added behind the reactions
        for c in at["centers"]:
            eng.atrp_add_center(**c)
        eng.atrp_init(at["interval"], at["num_particles"], at["ratio_activator"], at["ratio_deactivator"], at["delta_catalyst"],
                      at["k_activate"], at["k_deactivate"], select_from_all=at.get("select_from_all", True), seed=at.get("seed", spec.get("seed", 0)))
    return handles
