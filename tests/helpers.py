"""Shared test plumbing: build small systems on any Engine (oracle or HIP)."""
import numpy as np


def setup_small(eng, pos, types=None, box=20.0, rc=2.5, skin=0.3, dt=0.005, mass=1.0, vel=None,
                state=None, res_id=None, ids=None):
    pos = np.asarray(pos, dtype=np.float64)
    n = len(pos)
    eng.set_box([box] * 3 if np.isscalar(box) else box)
    eng.set_cutoff(rc, skin)
    eng.set_dt(dt)
    ids = np.arange(1, n + 1) if ids is None else ids
    types = np.zeros(n, np.int32) if types is None else types
    eng.set_particles(ids, types, pos, np.full(n, mass) if np.isscalar(mass) else mass, vel=vel,
                      state=state, res_id=res_id)
    return eng


def forces_energy(eng):
    """Forces (by id) and observables of the current configuration."""
    eng.run(0)
    return eng.get_state("FORCE"), eng.observe()


def total_epot(obs):
    return obs["epot_lj"] + obs["epot_tab"] + sum(obs["epot_list"])


def fd_forces(make, build, pos, h=1e-6):
    """-dU/dx by central differences; `build(eng, pos)` sets up the system."""
    pos = np.asarray(pos, dtype=np.float64)
    f = np.zeros_like(pos)
    for i in range(pos.shape[0]):
        for d in range(3):
            e = []
            for s in (+1, -1):
                p = pos.copy()
                p[i, d] += s * h
                eng = make()
                build(eng, p)
                eng.run(0)
                e.append(total_epot(eng.observe()))
                eng.close()
            f[i, d] = -(e[0] - e[1]) / (2 * h)
    return f


def sorted_events(ev):
    return [(int(e["step"]), int(e["id_a"]), int(e["id_b"]), int(e["reaction"]), float(e["r2"])) for e in ev]
