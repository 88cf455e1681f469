"""Engine -- thin object wrapper over one C-ABI context (include/chem_mi355.h).

One Engine == one `chem_ctx` == one GPU.  All arrays cross the boundary as plain
pointers + sizes (numpy buffers); nothing here computes physics.
"""
import ctypes as C

import numpy as np

from . import _capi


class ChemError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("chem error %d: %s" % (code, msg))
        self.code = code


def _ptr(a, typ):
    return a.ctypes.data_as(C.POINTER(typ)) if a is not None else None


def comm_unique_id():
    """RCCL unique id (128 bytes) for chem_comm_init; call on rank 0 and broadcast."""
    api = _capi.load()
    buf = C.create_string_buffer(128)
    rc = api.comm_unique_id(buf)
    if rc < 0:
        raise ChemError(rc, (api.last_error(None) or b"").decode())
    return buf.raw


class Engine:
    def __init__(self, device=0, precision=32, api=None, ctx=None):
        """precision: 32 (fp32 arrays, fp64 bonded/reaction distance) or 64 (all fp64)."""
        if api is None:
            api = _capi.load()
            ctx = api.create(int(device), int(precision))
            if not ctx:
                raise ChemError(_capi.EDEVICE, (api.last_error(None) or b"chem_create failed").decode())
        self.api, self.ctx = api, ctx
        self.precision = precision
        self._lists = {}  # handle -> arity

    def close(self):
        if self.ctx:
            self.api.destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise ChemError(rc, (self.api.last_error(self.ctx) or b"").decode())
        return rc

    # ---- set-up ----
    def set_box(self, L):
        L = np.ascontiguousarray(L, dtype=np.float64)
        self._ck(self.api.set_box(self.ctx, _ptr(L, C.c_double)))
        self.box = L.copy()

    def set_cutoff(self, max_cutoff, skin):
        self._ck(self.api.set_cutoff(self.ctx, float(max_cutoff), float(skin)))

    def set_dt(self, dt):
        self._ck(self.api.set_dt(self.ctx, float(dt)))

    def set_particles(self, ids, types, pos, mass, vel=None, q=None, state=None, res_id=None):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        n = ids.shape[0]
        types = np.ascontiguousarray(types, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(n, 3)
        mass = np.ascontiguousarray(np.broadcast_to(np.asarray(mass, dtype=np.float64), (n,)))
        vel = None if vel is None else np.ascontiguousarray(vel, dtype=np.float64).reshape(n, 3)
        q = None if q is None else np.ascontiguousarray(q, dtype=np.float64)
        state = None if state is None else np.ascontiguousarray(state, dtype=np.int32)
        res_id = None if res_id is None else np.ascontiguousarray(res_id, dtype=np.int32)
        self._ck(self.api.set_particles(
            self.ctx, n, _ptr(ids, C.c_int64), _ptr(types, C.c_int32), _ptr(pos, C.c_double),
            _ptr(vel, C.c_double), _ptr(mass, C.c_double), _ptr(q, C.c_double),
            _ptr(state, C.c_int32), _ptr(res_id, C.c_int32)))

    def modify_particle(self, pid, what, value):
        self._ck(self.api.modify_particle(self.ctx, int(pid), _capi.STATE[what.upper()], float(value)))

    def set_exclusions(self, pairs):
        p = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
        self._ck(self.api.set_exclusions(self.ctx, p.shape[0], _ptr(p, C.c_int64)))

    def nb_lj(self, t1, t2, eps, sig, rc, shift_auto=True):
        self._ck(self.api.nb_lj(self.ctx, t1, t2, eps, sig, rc, 1 if shift_auto else 0))

    def nb_table(self, t1, t2, r0, dr, e, f, rc):
        e = np.ascontiguousarray(e, dtype=np.float64)
        f = np.ascontiguousarray(f, dtype=np.float64)
        assert e.shape == f.shape
        self._ck(self.api.nb_table(self.ctx, t1, t2, e.shape[0], r0, dr, _ptr(e, C.c_double),
                                   _ptr(f, C.c_double), rc))

    def list_create(self, arity, kind, by_types=False):
        kind = _capi.POT[kind] if isinstance(kind, str) else kind
        h = self._ck(self.api.list_create(self.ctx, arity, kind, 1 if by_types else 0))
        self._lists[h] = arity
        return h

    def list_add(self, h, ids):
        a = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1, self._lists[h])
        if a.shape[0]:
            self._ck(self.api.list_add(self.ctx, h, a.shape[0], _ptr(a, C.c_int64)))

    def list_set_params(self, h, params, types=None):
        p = np.ascontiguousarray(params, dtype=np.float64)
        t = list(types) if types is not None else []
        t = t + [-1] * (4 - len(t))
        self._ck(self.api.list_set_params(self.ctx, h, t[0], t[1], t[2], t[3], _ptr(p, C.c_double), p.shape[0]))

    def get_list(self, h):
        n = self._ck(self.api.get_list(self.ctx, h, None, 0))
        out = np.empty((n, self._lists[h]), dtype=np.int64)
        if n:
            self._ck(self.api.get_list(self.ctx, h, _ptr(out, C.c_int64), n))
        return out

    def thermostat_langevin(self, kT, gamma, seed=0):
        self._ck(self.api.thermostat_langevin(self.ctx, float(kT), float(gamma), int(seed)))

    def thermostat_langevin_types(self, types):
        """Thermal groups (LangevinThermostat.add_valid_types): thermalise only these particle types; [] = all."""
        t = np.ascontiguousarray(list(types), dtype=np.int32)
        self._ck(self.api.thermostat_langevin_types(self.ctx, t.shape[0], _ptr(t, C.c_int32)))

    def table_create(self, r0, dr, e, f):
        """Bond table (rows of a .pot file); returns the handle to pass as list_set_params(h, [handle])."""
        e = np.ascontiguousarray(e, dtype=np.float64); f = np.ascontiguousarray(f, dtype=np.float64)
        assert e.shape == f.shape
        return self._ck(self.api.table_create(self.ctx, e.shape[0], float(r0), float(dr), _ptr(e, C.c_double), _ptr(f, C.c_double)))

    def thermostat_rescale(self, kind, kT, param):
        """kind: 'berendsen' (param = tau), 'isokinetic' (param = coupling steps) or None/0 (off)."""
        k = {None: 0, 0: 0, "berendsen": 1, 1: 1, "isokinetic": 2, 2: 2}[kind]
        self._ck(self.api.thermostat_rescale(self.ctx, k, float(kT), float(param)))

    def thermostat_svr(self, kT, coupling, seed=0):
        """StochasticVelocityRescaling: kT, coupling time (same units as dt); coupling <= 0 switches it off."""
        self._ck(self.api.thermostat_svr(self.ctx, float(kT), float(coupling), int(seed)))

    def cap_force(self, max_force):
        self._ck(self.api.cap_force(self.ctx, float(max_force)))

    def reaction_init(self, interval, nearest=True, max_per_interval=0, seed=0):
        self._ck(self.api.reaction_init(self.ctx, int(interval), 1 if nearest else 0, int(max_per_interval), int(seed)))

    def reaction_add(self, type_1, type_2, delta_1, delta_2, min_state_1, max_state_1, min_state_2,
                     max_state_2, rate, cutoff, bond_list=-1, min_cutoff=0.0, intramolecular=False,
                     intraresidual=False, is_virtual=False, active=True, new_type_1=-1, new_type_2=-1,
                     new_mass_1=0.0, new_mass_2=0.0, new_q_1=0.0, new_q_2=0.0):
        d = _capi.ReactionDesc(type_1, type_2, delta_1, delta_2, min_state_1, max_state_1, min_state_2,
                               max_state_2, rate, cutoff, min_cutoff, int(bool(intramolecular)),
                               int(bool(intraresidual)), int(bool(is_virtual)), int(bool(active)),
                               bond_list, new_type_1, new_type_2, new_mass_1, new_mass_2, new_q_1, new_q_2)
        return self._ck(self.api.reaction_add(self.ctx, C.byref(d)))

    def reaction_neighbour_change(self, reaction, invoke_on, old_type, nb_level, new_type, new_mass, new_q=0.0, new_state=None,
                                  incr_state=None, state_window=None):
        """PostProcessChangeNeighboursProperty rule for the events of `reaction` (index from reaction_add):
        invoke_on 'type_1' | 'type_2' | 'both'; new_state None keeps the chemical state, incr_state adds to it;
        state_window (min, max): only neighbours whose state is in [min, max)."""
        io = {"type_1": 1, "type_2": 2, "both": 3, 1: 1, 2: 2, 3: 3}[invoke_on]
        mode, val = (2, int(incr_state)) if incr_state is not None else ((1, int(new_state)) if new_state is not None else (0, 0))
        lo, hi = (0, 0) if state_window is None else (int(state_window[0]), int(state_window[1]))
        r = _capi.NbChange(int(reaction), io, int(old_type), int(nb_level), int(new_type), mode, val, 0, float(new_mass), float(new_q), lo, hi)
        self._ck(self.api.reaction_neighbour_change(self.ctx, C.byref(r)))

    def reaction_constraint(self, reaction, role, nb_type, min_state, max_state):
        """ReactionConstraintNeighbourState on role 'type_1' | 'type_2' of `reaction` (reaction_setup.py:203-204)."""
        ro = {"type_1": 1, "type_2": 2, 1: 1, 2: 2}[role]
        self._ck(self.api.reaction_constraint(self.ctx, int(reaction), ro, int(nb_type), int(min_state), int(max_state)))

    def reaction_restrict(self, reaction, id_pairs):
        """RestrictReaction.define_connection: only these unordered id pairs may react in `reaction` (index from reaction_add)."""
        p = np.ascontiguousarray(id_pairs, dtype=np.int64).reshape(-1, 2)
        self._ck(self.api.reaction_restrict(self.ctx, int(reaction), p.shape[0], _ptr(p, C.c_int64)))

    def atrp_init(self, interval, num_particles, ratio_activator, ratio_deactivator, delta_catalyst, k_activate, k_deactivate,
                  select_from_all=True, seed=0):
        """integrator.ATRPActivator(system, interval, num_particles, ...) (reaction_post_process.py:380-426)."""
        d = _capi.AtrpDesc(int(interval), int(num_particles), 1 if select_from_all else 0, 0, float(ratio_activator), float(ratio_deactivator),
                           float(delta_catalyst), float(k_activate), float(k_deactivate), int(seed))
        self._ck(self.api.atrp_init(self.ctx, C.byref(d)))

    def atrp_disconnect(self):
        self._ck(self.api.atrp_init(self.ctx, None))

    def atrp_add_center(self, type_id, state, is_activator, new_type, new_mass, new_q=0.0, delta_state=0):
        self._ck(self.api.atrp_add_center(self.ctx, int(type_id), int(state), 1 if is_activator else 0, int(new_type), float(new_mass), float(new_q), int(delta_state)))

    def atrp_stats(self):
        n = self._ck(self.api.atrp_get_stats(self.ctx, None, 0))
        buf = (_capi.AtrpStats * max(n, 1))()
        if n:
            self._ck(self.api.atrp_get_stats(self.ctx, buf, n))
        return [{k: getattr(buf[i], k) for k, _ in _capi.AtrpStats._fields_} for i in range(n)]

    def topology_register(self, h, types):
        t = np.ascontiguousarray(types, dtype=np.int32)
        self._ck(self.api.topology_register(self.ctx, t.shape[0], h, _ptr(t, C.c_int32)))

    def reactions_enable(self, on=True):
        self._ck(self.api.reactions_enable(self.ctx, 1 if on else 0))

    def reaction_set_rate(self, r, rate):
        self._ck(self.api.reaction_set_rate(self.ctx, r, float(rate)))

    # ---- hot call ----
    def run(self, nsteps):
        self._ck(self.api.run(self.ctx, int(nsteps)))

    # ---- read-back ----
    @property
    def n(self):
        return self.api.num_particles(self.ctx)

    @property
    def step(self):
        return self.api.get_step(self.ctx)

    def get_state(self, what):
        w = _capi.STATE[what.upper()]
        n = self.n
        if what.upper() in ("POS", "VEL", "FORCE", "POS_UNFOLDED"):
            out = np.empty((n, 3), dtype=np.float64)
        elif what.upper() == "IMAGE":
            out = np.empty((n, 3), dtype=np.int32)
        elif what.upper() == "MASS":
            out = np.empty(n, dtype=np.float64)
        elif what.upper() == "ID":
            out = np.empty(n, dtype=np.int64)
        else:
            out = np.empty(n, dtype=np.int32)
        self._ck(self.api.get_state(self.ctx, w, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def get_events(self):
        n = self._ck(self.api.get_events(self.ctx, None, 0))
        buf = (_capi.Event * max(n, 1))()
        if n:
            self._ck(self.api.get_events(self.ctx, buf, n))
        dt = np.dtype([("step", "i8"), ("id_a", "i8"), ("id_b", "i8"), ("reaction", "i4"), ("pad", "i4"), ("r2", "f8")])
        return np.frombuffer(buf, dtype=dt, count=n).copy()

    def get_exclusions(self):
        n = self._ck(self.api.get_exclusions(self.ctx, None, 0))
        out = np.empty((n, 2), dtype=np.int64)
        if n:
            self._ck(self.api.get_exclusions(self.ctx, _ptr(out, C.c_int64), n))
        return out

    def get_verlet_pairs(self):
        n = self._ck(self.api.get_verlet_pairs(self.ctx, None, 0))
        out = np.empty((n, 2), dtype=np.int64)
        if n:
            self._ck(self.api.get_verlet_pairs(self.ctx, _ptr(out, C.c_int64), n))
        return out

    def debug_force_list(self, tag):
        """Diagnostic (tests): partner tags of the 16-bit force list of particle `tag`, in list order."""
        lib = self.api.lib
        lib.chem_debug_force_list.restype = C.c_int64
        buf = np.zeros(1024, dtype=np.int32)
        m = lib.chem_debug_force_list(C.c_void_p(self.ctx), C.c_int32(int(tag)), buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.size))
        if m < 0:
            raise ChemError(int(m), "force list not available (no LDS tiles)")
        return buf[:m].copy()

    def observe(self):
        o = _capi.Obs()
        self._ck(self.api.observe(self.ctx, C.byref(o)))
        nl = len(self._lists)
        return dict(step=o.step, npart=o.npart, ekin=o.ekin, temperature=o.temperature,
                    epot_lj=o.epot_lj, epot_tab=o.epot_tab, epot_list=list(o.epot_list)[:nl],
                    list_size=list(o.list_size)[:nl], momentum=list(o.momentum), virial_nb=o.virial_nb)

    def timers(self):
        t = _capi.Timers()
        self._ck(self.api.get_timers(self.ctx, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _capi.Timers._fields_}

    # ---- product-only knobs ----
    def set_option(self, name, value):
        self._ck(self.api.set_option(self.ctx, name.encode(), float(value)))

    def set_nlist_capacity(self, n):
        self._ck(self.api.set_nlist_capacity(self.ctx, int(n)))

    def comm_init(self, nranks, rank, uid):
        """Join the slab decomposition (node grid (1,1,nranks)); uid from comm_unique_id() on rank 0."""
        grid = (C.c_int * 3)(1, 1, nranks)
        self._ck(self.api.comm_init(self.ctx, nranks, rank, grid, uid))

    def comm_init_ipc(self, nranks, rank, shm_name):
        """One process per rank over hipIpc handles (ranks may share a device); shm_name: same fresh name on all ranks."""
        self._ck(self.api.comm_init_ipc(self.ctx, nranks, rank, shm_name.encode()))

    def comm_init_local(self, nranks, rank, hub_id):
        self._ck(self.api.comm_init_local(self.ctx, nranks, rank, hub_id))

    def sync(self):
        self._ck(self.api.device_sync(self.ctx))
