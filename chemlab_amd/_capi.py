"""ctypes binding of the C ABI declared in include/chem_mi355.h.

`load()` returns the bound product library (libchem_mi355.so, HIP/gfx950).  There is no
CPU fall-back: if the shared object is missing or cannot be loaded this raises
`ChemLibraryError` -- build it with `python -c "import __graft_entry__ as g; g.build()"`
(or `make -C chemlab_amd/csrc`).

`bind(lib, prefix)` attaches the signatures to any library exporting the same entry
points under `prefix`; tests use it to drive the CPU oracle (prefix `orc_`) through the
same host code.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHEM_MI355_LIB") or os.path.join(HERE, "csrc", "libchem_mi355.so")   # override: A/B builds of the same source (tools/)

CHEM_MAX_LISTS = 32
CHEM_MAX_TYPES = 16
CHEM_MAX_POT_PARAMS = 6

# error codes
OK, EINVAL, ENOSPC, EDEVICE, ESTATE, ENOTIMPL, ECOMM = 0, -1, -2, -3, -4, -5, -6
PREC_F32, PREC_F64 = 32, 64

POT = dict(HARMONIC=1, FENE=2, TABULATED=3, FENE_LJ=4, LJ_BOND=5, DIH_HARMONIC=23, ANG_HARMONIC=10, ANG_COSINE=11, ANG_TABULATED=12, DIH_NCOS=20, DIH_RB=21, DIH_TABULATED=22)
STATE = dict(POS=1, VEL=2, FORCE=3, TYPE=4, STATE=5, RESID=6, MASS=7, ID=8, IMAGE=9, MOLID=10,
             POS_UNFOLDED=11)


class ChemLibraryError(RuntimeError):
    pass


class ReactionDesc(C.Structure):
    _fields_ = [
        ("type_1", C.c_int32), ("type_2", C.c_int32),
        ("delta_1", C.c_int32), ("delta_2", C.c_int32),
        ("min_state_1", C.c_int32), ("max_state_1", C.c_int32),
        ("min_state_2", C.c_int32), ("max_state_2", C.c_int32),
        ("rate", C.c_double), ("cutoff", C.c_double), ("min_cutoff", C.c_double),
        ("intramolecular", C.c_int32), ("intraresidual", C.c_int32),
        ("is_virtual", C.c_int32), ("active", C.c_int32),
        ("bond_list", C.c_int32),
        ("new_type_1", C.c_int32), ("new_type_2", C.c_int32),
        ("new_mass_1", C.c_double), ("new_mass_2", C.c_double),
        ("new_q_1", C.c_double), ("new_q_2", C.c_double),
    ]


class NbChange(C.Structure):
    _fields_ = [("reaction", C.c_int32), ("invoke_on", C.c_int32), ("old_type", C.c_int32), ("nb_level", C.c_int32),
                ("new_type", C.c_int32), ("set_state", C.c_int32), ("new_state", C.c_int32), ("pad", C.c_int32),
                ("new_mass", C.c_double), ("new_q", C.c_double), ("min_state", C.c_int32), ("max_state", C.c_int32)]


class AtrpDesc(C.Structure):
    _fields_ = [("interval", C.c_int32), ("num_particles", C.c_int32), ("select_from_all", C.c_int32), ("pad", C.c_int32),
                ("ratio_activator", C.c_double), ("ratio_deactivator", C.c_double), ("delta_catalyst", C.c_double),
                ("k_activate", C.c_double), ("k_deactivate", C.c_double), ("seed", C.c_uint64)]


class AtrpStats(C.Structure):
    _fields_ = [("step", C.c_int64), ("candidates", C.c_int64), ("selected", C.c_int64), ("activated", C.c_int64),
                ("deactivated", C.c_int64), ("ratio_activator", C.c_double), ("ratio_deactivator", C.c_double)]


class Event(C.Structure):
    _fields_ = [("step", C.c_int64), ("id_a", C.c_int64), ("id_b", C.c_int64),
                ("reaction", C.c_int32), ("pad", C.c_int32), ("r2", C.c_double)]


class Obs(C.Structure):
    _fields_ = [("step", C.c_int64), ("npart", C.c_int64), ("ekin", C.c_double),
                ("temperature", C.c_double), ("epot_lj", C.c_double), ("epot_tab", C.c_double),
                ("epot_list", C.c_double * CHEM_MAX_LISTS), ("list_size", C.c_int64 * CHEM_MAX_LISTS),
                ("momentum", C.c_double * 3), ("virial_nb", C.c_double)]


class Timers(C.Structure):
    _fields_ = [("run_wall_s", C.c_double), ("steps", C.c_int64), ("rebuilds", C.c_int64),
                ("reaction_steps", C.c_int64), ("nlist_entries", C.c_int64),
                ("nlist_capacity", C.c_int64), ("reaction_wall_s", C.c_double),
                ("rebuild_wall_s", C.c_double), ("pair_kernel_ms", C.c_double),
                ("pair_kernel_launches", C.c_int64),
                ("rebuild_kernel_ms", C.c_double), ("rebuild_kernel_launches", C.c_int64),
                ("decide_kernel_ms", C.c_double), ("decide_kernel_launches", C.c_int64),
                ("nlist_entries_all", C.c_int64),
                ("integrate_kernel_ms", C.c_double), ("integrate_kernel_launches", C.c_int64),
                ("bonded_kernel_ms", C.c_double), ("bonded_kernel_launches", C.c_int64), ("list_rebuilds", C.c_int64)]


_P = C.c_void_p
_i, _i64, _d, _u64 = C.c_int, C.c_int64, C.c_double, C.c_uint64
_pd, _pi32, _pi64 = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)

# name -> (restype, argtypes); ctx-taking functions list the ctx as first argument.
SIGNATURES = {
    "destroy": (None, [_P]),
    "last_error": (C.c_char_p, [_P]),
    "set_box": (_i, [_P, _pd]),
    "set_cutoff": (_i, [_P, _d, _d]),
    "set_dt": (_i, [_P, _d]),
    "set_particles": (_i, [_P, _i64, _pi64, _pi32, _pd, _pd, _pd, _pd, _pi32, _pi32]),
    "modify_particle": (_i, [_P, _i64, _i, _d]),
    "set_exclusions": (_i, [_P, _i64, _pi64]),
    "nb_lj": (_i, [_P, _i, _i, _d, _d, _d, _i]),
    "nb_table": (_i, [_P, _i, _i, _i64, _d, _d, _pd, _pd, _d]),
    "list_create": (_i, [_P, _i, _i, _i]),
    "list_add": (_i, [_P, _i, _i64, _pi64]),
    "list_set_params": (_i, [_P, _i, _i, _i, _i, _i, _pd, _i]),
    "get_list": (_i64, [_P, _i, _pi64, _i64]),
    "thermostat_langevin": (_i, [_P, _d, _d, _u64]),
    "thermostat_langevin_types": (_i, [_P, _i, _pi32]),
    "cap_force": (_i, [_P, _d]),
    "thermostat_rescale": (_i, [_P, _i, _d, _d]),
    "thermostat_svr": (_i, [_P, _d, _d, _u64]),
    "table_create": (_i, [_P, C.c_int64, _d, _d, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "reaction_init": (_i, [_P, _i, _i, _i, _u64]),
    "reaction_add": (_i, [_P, C.POINTER(ReactionDesc)]),
    "reaction_neighbour_change": (_i, [_P, C.POINTER(NbChange)]),
    "reaction_restrict": (_i, [_P, _i, _i64, _pi64]),
    "reaction_constraint": (_i, [_P, _i, _i, _i, _i, _i]),
    "atrp_init": (_i, [_P, C.POINTER(AtrpDesc)]),
    "atrp_add_center": (_i, [_P, _i, _i, _i, _i, _d, _d, _i]),
    "atrp_get_stats": (_i64, [_P, C.POINTER(AtrpStats), _i64]),
    "topology_register": (_i, [_P, _i, _i, _pi32]),
    "reactions_enable": (_i, [_P, _i]),
    "reaction_set_rate": (_i, [_P, _i, _d]),
    "run": (_i, [_P, _i64]),
    "num_particles": (_i64, [_P]),
    "get_step": (_i64, [_P]),
    "get_state": (_i64, [_P, _i, _P, _i64]),
    "get_events": (_i64, [_P, C.POINTER(Event), _i64]),
    "get_exclusions": (_i64, [_P, _pi64, _i64]),
    "get_verlet_pairs": (_i64, [_P, _pi64, _i64]),
    "observe": (_i, [_P, C.POINTER(Obs)]),
    "get_timers": (_i, [_P, C.POINTER(Timers)]),
    "set_option": (_i, [_P, C.c_char_p, _d]),
}

# entry points that only the product library exports
PRODUCT_ONLY = {
    "create": (_P, [_i, _i]),
    "abi_version": (_i, []),
    "device_sync": (_i, [_P]),
    "set_nlist_capacity": (_i, [_P, _i]),
    "comm_unique_id": (_i, [C.c_char_p]),
    "comm_init": (_i, [_P, _i, _i, C.POINTER(C.c_int), C.c_char_p]),
    "comm_init_local": (_i, [_P, _i, _i, _i]),
    "comm_init_ipc": (_i, [_P, _i, _i, C.c_char_p]),
}


class Api:
    """Namespace of bound functions: api.run(ctx, n) == <prefix>run(ctx, n)."""

    def __init__(self, lib, prefix, extra=None):
        self.lib, self.prefix = lib, prefix
        table = dict(SIGNATURES)
        table.update(extra or {})
        for name, (res, args) in table.items():
            fn = getattr(lib, prefix + name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)

    def exported(self):
        return sorted(k for k in self.__dict__ if k not in ("lib", "prefix"))


def bind(lib, prefix, extra=None):
    return Api(lib, prefix, extra)


_product = None


def load():
    """Load libchem_mi355.so (the HIP product path).  Fails loudly when it is absent."""
    global _product
    if _product is None:
        if not os.path.exists(LIB_PATH):
            raise ChemLibraryError(
                "HIP extension %s not built; there is no CPU fall-back. "
                "Run `make -C chemlab_amd/csrc` (hipcc --offload-arch=gfx950)." % LIB_PATH)
        try:
            lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        except OSError as e:  # missing ROCm runtime etc.
            raise ChemLibraryError("cannot load %s: %s" % (LIB_PATH, e))
        _product = bind(lib, "chem_", PRODUCT_ONLY)
    return _product


def header_symbols():
    """Every chem_* function the public header declares (used by the CPU symbol test)."""
    import re
    hdr = os.path.join(HERE, "..", "include", "chem_mi355.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chem_[a-z0-9_]+)\s*\(", text)))
