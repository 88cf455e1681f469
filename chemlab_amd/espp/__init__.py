"""py3 `espressopp`-shaped shim over the C ABI (include/chem_mi355.h).

ChemLab reaches its hot path through Boost.Python objects of the external module `espressopp`
(`import espressopp`, /root/reference/src/start_simulation.py:20).  This module exposes the subset
of that surface the in-scope driver logic touches, with the same names, argument
meaning and error behaviour, implemented on one `chemlab_amd.engine.Engine` per `System`.
Everything outside the hot-path scope raises NotImplementedError naming the symbol.

    import chemlab_amd.espp as espressopp
    system = espressopp.System()
    ...
    integrator.run(n)          # -> chem_run(ctx, n)

The engine factory is replaceable (`set_engine_factory`) so that tests can drive the very same
driver code with the CPU oracle.
"""
import math
import types

import numpy as np

from ..engine import Engine
from .. import rank as _rank

_factory = [lambda: Engine(device=0, precision=32)]


def set_engine_factory(fn):
    """fn() -> Engine-like object (tests: the CPU oracle)."""
    _factory[0] = fn


class Real3D(object):
    def __init__(self, x=0.0, y=None, z=None):
        if y is None:
            v = list(x) if hasattr(x, "__len__") else [x, x, x]
            x, y, z = v
        self.v = np.array([x, y, z], dtype=np.float64)

    def __getitem__(self, i):
        return self.v[i]

    def __len__(self):
        return 3

    def __iter__(self):
        return iter(self.v)

    def __mul__(self, s):
        return Real3D(*(self.v * s))

    def __repr__(self):
        return "Real3D(%g, %g, %g)" % tuple(self.v)


class Int3D(Real3D):
    pass


def _unsupported(name):
    def ctor(*a, **k):
        raise NotImplementedError("espressopp.%s is outside the MI355X hot-path scope (SURVEY.md 8b)" % name)
    return ctor


# --------------------------------------------------------------------------------------------
class System(object):
    """espressopp.System(): .rng .skin .bc .storage .integrator (start_simulation.py:148-163)."""

    def __init__(self):
        self.engine = _factory[0]()
        self.rng = None
        self.skin = 0.0
        self.bc = None
        self.storage = None
        self.integrator = None
        self.topology_manager = None
        self._interactions = []     # (interaction, name)
        self.max_cutoff = None

    def addInteraction(self, interaction, name=None):
        self._interactions.append((interaction, name if name is not None else "interaction_%d" % len(self._interactions)))

    def getAllInteractions(self):
        return {name: i for i, name in self._interactions}

    def getNumberOfInteractions(self):
        return len(self._interactions)

    def getInteraction(self, k):
        return self._interactions[k][0]

    def getNameOfInteraction(self, k):
        return self._interactions[k][1]


class _RNG(object):
    def __init__(self, seed=0):
        self._seed = int(seed)

    def seed(self, s):
        self._seed = int(s)

    def get_seed(self):
        return self._seed


class _OrthorhombicBC(object):
    def __init__(self, rng, boxL):
        self.rng = rng
        self.boxL = Real3D(*[float(b) for b in boxL])


class _Particle(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _DomainDecomposition(object):
    """storage.DomainDecomposition(system, nodeGrid, cellGrid): addParticles/decompose/getParticle/
    modifyParticle (start_simulation.py:163-171; examples/atrp_lj/hooks.py:63-65)."""

    def __init__(self, system, nodeGrid=None, cellGrid=None):
        self.system = system
        self.nodeGrid, self.cellGrid = nodeGrid, cellGrid
        self._pending = []
        self._props = None
        self._uploaded = False
        system.engine.set_box(list(system.bc.boxL))

    def addParticles(self, plist, *props):
        self._props = list(props)
        self._pending.extend(plist)

    def addParticle(self, pid, pos):
        self._props = ["id", "pos"]
        self._pending.append([pid, pos])

    def decompose(self):
        if self._uploaded or not self._pending:
            return
        pr = self._props
        col = lambda name, default=None: [p[pr.index(name)] for p in self._pending] if name in pr else default
        n = len(self._pending)
        ids = np.array(col("id"), dtype=np.int64)
        pos = np.array([list(p) for p in col("pos")], dtype=np.float64)
        types = np.array(col("type", [0] * n), dtype=np.int32)
        mass = np.array(col("mass", [1.0] * n), dtype=np.float64)
        q = np.array(col("q", [0.0] * n), dtype=np.float64)
        res_id = np.array(col("res_id", list(ids)), dtype=np.int32)
        state = np.array(col("state", [0] * n), dtype=np.int32)
        vel = None
        if "v" in pr:
            vel = np.array([list(p) for p in col("v")], dtype=np.float64)
        self.system.engine.set_particles(ids, types, pos, mass, vel=vel, q=q, state=state, res_id=res_id)
        self._uploaded = True
        self._ids = ids
        self._pending = []

    def particleExists(self, pid):
        return bool(np.any(self._ids == pid))

    def getParticle(self, pid):
        e = self.system.engine
        k = int(np.searchsorted(np.sort(self._ids), pid))
        g = lambda w: e.get_state(w)[k]
        return _Particle(id=pid, pos=Real3D(*g("POS")), v=Real3D(*g("VEL")), f=Real3D(*g("FORCE")), type=int(g("TYPE")),
                         mass=float(g("MASS")), state=int(g("STATE")), res_id=int(g("RESID")), imageBox=Int3D(*g("IMAGE")), q=0.0)

    def modifyParticle(self, pid, prop, value):
        self.system.engine.modify_particle(pid, {"type": "TYPE", "state": "STATE", "mass": "MASS", "res_id": "RESID"}[prop], value)


def _node_grid(n, *a, **k):
    """tools.decomp.nodeGrid(MPI size): this build decomposes along z only (multigpu.node_grid)."""
    return Int3D(1, 1, int(n))


def _cell_grid(box, node_grid, rc, skin, *a, **k):
    """tools.decomp.cellGrid(box, nodeGrid, max_cutoff, skin): cells of edge >= rc+skin per node;
    known answer examples/atrp_lj/single:39 -> (6, 6, 6) for box 13.40248, rc 2.0, skin 0.1."""
    rl = rc + skin
    ng = list(node_grid)
    g = [int(math.floor(float(box[d]) / (rl * ng[d]))) for d in range(3)]
    if min(g) < 1:
        raise RuntimeError("local box too small for the cutoff + skin")
    return Int3D(*g)


class DynamicExcludeList(object):
    """espressopp.DynamicExcludeList(integrator, pairs) (start_simulation.py:189); the device list
    observes every Fixed*List by construction, observe_* only record the request."""

    def __init__(self, integrator, exclusions=None):
        self.system = integrator.system
        self.exclusions = [tuple(p) for p in (exclusions or [])]
        if self.exclusions:
            self.system.engine.set_exclusions(self.exclusions)
        self.observed = []

    def exclude(self, a, b):
        self.exclusions.append((a, b))
        self.system.engine.set_exclusions(self.exclusions)

    def observe_tuple(self, fpl):
        self.observed.append(fpl)

    observe_triple = observe_quadruple = observe_tuple

    def get_list(self):
        return [tuple(p) for p in self.system.engine.get_exclusions().tolist()]

    @property
    def size(self):
        return len(self.system.engine.get_exclusions())


class VerletList(object):
    """espressopp.VerletList(system, cutoff, exclusionlist) (start_simulation.py:193-197)."""

    def __init__(self, system, cutoff, exclusionlist=None):
        self.system, self.cutoff, self.exclusionlist = system, cutoff, exclusionlist
        system.max_cutoff = cutoff
        system.engine.set_cutoff(cutoff, system.skin)

    def totalSize(self):
        return len(self.system.engine.get_verlet_pairs())

    def get_timers(self):
        return []


# ---- fixed lists ---------------------------------------------------------------------------
class _FixedList(object):
    arity = 2

    def __init__(self, storage):
        self.system = storage.system
        self.handle = None
        self._pending = []

    def _add(self, entries):
        entries = [tuple(int(x) for x in e) for e in entries]
        if self.handle is None:
            self._pending.extend(entries)
        elif entries:
            self.system.engine.list_add(self.handle, entries)

    def _bind(self, kind, by_types):
        if self.handle is None:
            self.handle = self.system.engine.list_create(self.arity, kind, by_types)
            if self._pending:
                self.system.engine.list_add(self.handle, self._pending)
            self._pending = []
        return self.handle

    def _all(self):
        if self.handle is None:
            return list(self._pending)
        return [tuple(r) for r in self.system.engine.get_list(self.handle).tolist()]

    def totalSize(self):
        return len(self._all())

    size = totalSize


class FixedPairList(_FixedList):
    arity = 2

    def addBonds(self, bonds):
        self._add(bonds)

    def add(self, a, b):
        self._add([(a, b)])

    def getAllBonds(self):
        return self._all()

    getBonds = getAllBonds


class FixedTripleList(_FixedList):
    arity = 3

    def addTriples(self, t):
        self._add(t)

    def getAllTriples(self):
        return self._all()


class FixedQuadrupleList(_FixedList):
    arity = 4

    def addQuadruples(self, t):
        self._add(t)

    def getAllQuadruples(self):
        return self._all()


# ---- potentials (parameter holders) -----------------------------------------------------------
class _Pot(object):
    kind = None

    def params(self):
        raise NotImplementedError


class _LennardJones(_Pot):
    def __init__(self, epsilon=1.0, sigma=1.0, cutoff=2.5, shift="auto"):
        self.epsilon, self.sigma, self.cutoff, self.shift = epsilon, sigma, cutoff, shift


class _Tabulated(_Pot):
    """interaction.Tabulated(itype, filename, cutoff): rows `r e f` of an ESPResSo++ .pot file."""

    def __init__(self, itype=1, filename=None, cutoff=None):
        if itype != 1:
            raise NotImplementedError("Tabulated itype %r (only linear interpolation, itype=1, is in scope)" % itype)
        self.itype, self.filename, self.cutoff = itype, filename, cutoff
        tab = np.loadtxt(filename)
        self.r, self.e, self.f = tab[:, 0], tab[:, 1], tab[:, 2]
        self.r0, self.dr = float(self.r[0]), float(self.r[1] - self.r[0])


class _MixedTabulated(_Pot):
    """interaction.MixedTabulated(itype, table1, table2, chemical_conversion=None, cutoff=None, mix_value=None)
    (gromacs_topology.py:756-790, nonbond_params func 10 / 12): U = x * table1 + (1 - x) * table2 with x the value of a
    ChemicalConversion observable (func 10: followed whenever the observable is computed) or a constant (func 12).
    Linear interpolation is linear in the rows, so the mixture IS one table: x * rows1 + (1 - x) * rows2 -- re-sent to the
    engine (chem_nb_table) when x changes; the kernels see an ordinary tabulated pair potential."""

    def __init__(self, itype=1, table1=None, table2=None, chemical_conversion=None, cutoff=None, mix_value=None):
        if itype != 1:
            raise NotImplementedError("MixedTabulated itype %r (only linear interpolation, itype=1, is in scope)" % itype)
        a, b = np.loadtxt(table1), np.loadtxt(table2)
        if a.shape != b.shape or not np.allclose(a[:, 0], b[:, 0], rtol=0, atol=1e-12):
            raise ValueError("MixedTabulated: %s and %s are not on the same r grid" % (table1, table2))
        self.table1, self.table2, self.cutoff = table1, table2, cutoff
        self.r, self._e, self._f = a[:, 0], (a[:, 1], b[:, 1]), (a[:, 2], b[:, 2])
        self.r0, self.dr = float(self.r[0]), float(self.r[1] - self.r[0])
        self.observable = chemical_conversion
        self.mix_value = 0.0 if mix_value is None else float(mix_value)

    def rows(self, x):
        return x * self._e[0] + (1.0 - x) * self._e[1], x * self._f[0] + (1.0 - x) * self._f[1]


class _TabulatedAngular(_Tabulated):
    """interaction.TabulatedAngular(itype, filename): rows `theta U -dU/dtheta` (radians) of a table_a<N>.pot file."""


class _TabulatedDihedral(_Tabulated):
    """interaction.TabulatedDihedral(itype, filename): rows `phi U -dU/dphi` (radians, [-pi, pi]) of a table_d<N>.pot file."""


class _Harmonic(_Pot):
    kind = "HARMONIC"

    def __init__(self, K=1.0, r0=0.0, cutoff=None, shift=0.0):
        self.K, self.r0 = K, r0

    def params(self):
        return [self.K, self.r0]


class _FENE(_Pot):
    kind = "FENE"

    def __init__(self, K=1.0, r0=0.0, rMax=1.0, cutoff=None, shift=0.0):
        self.K, self.r0, self.rMax = K, r0, rMax

    def params(self):
        return [self.K, self.r0, self.rMax]


class _FENELennardJones(_Pot):
    """interaction.FENELennardJones(K, r0, rMax, sigma, epsilon) (bond func 9, gromacs_topology.py:935-942)."""
    kind = "FENE_LJ"

    def __init__(self, K=1.0, r0=0.0, rMax=1.0, sigma=1.0, epsilon=1.0, cutoff=None, shift=0.0):
        self.K, self.r0, self.rMax, self.sigma, self.epsilon = K, r0, rMax, sigma, epsilon

    def params(self):
        return [self.K, self.r0, self.rMax, self.sigma, self.epsilon]


class _LJBond(_Pot):
    """LennardJones as the potential of a FixedPairListLennardJones (1-4 pairs, gromacs_topology.py:1314-1411)."""
    kind = "LJ_BOND"


class _DihedralHarmonic(_Pot):
    """interaction.DihedralHarmonic(K, phi0) (dihedral func 12, gromacs_topology.py:1199-1202)."""
    kind = "DIH_HARMONIC"

    def __init__(self, K=0.0, phi0=0.0):
        self.K, self.phi0 = K, phi0

    def params(self):
        return [self.K, self.phi0]


class _AngularHarmonic(_Pot):
    kind = "ANG_HARMONIC"

    def __init__(self, K=1.0, theta0=0.0):
        self.K, self.theta0 = K, theta0

    def params(self):
        return [self.K, self.theta0]


class _Cosine(_Pot):
    kind = "ANG_COSINE"

    def __init__(self, K=1.0, theta0=0.0):
        self.K, self.theta0 = K, theta0

    def params(self):
        return [self.K, self.theta0]


class _DihedralHarmonicNCos(_Pot):
    kind = "DIH_NCOS"

    def __init__(self, K=0.0, phi0=0.0, multiplicity=1):
        self.K, self.phi0, self.n = K, phi0, multiplicity

    def params(self):
        return [self.K, self.phi0, float(self.n)]


class _DihedralRB(_Pot):
    kind = "DIH_RB"

    def __init__(self, K0=0.0, K1=0.0, K2=0.0, K3=0.0, K4=0.0, K5=0.0):
        self.C = [K0, K1, K2, K3, K4, K5]

    def params(self):
        return list(self.C)


# ---- interactions ----------------------------------------------------------------------------
class _VerletListInteraction(object):
    def __init__(self, vl):
        self.vl, self.system = vl, vl.system
        self._pots = {}

    def getPotential(self, t1, t2):
        return self._pots[(min(t1, t2), max(t1, t2))]

    def getAllPotentials(self):
        return dict(self._pots)


class _VerletListLennardJones(_VerletListInteraction):
    label = "lj"

    def setPotential(self, type1, type2, potential):
        self._pots[(min(type1, type2), max(type1, type2))] = potential
        self.system.engine.nb_lj(type1, type2, potential.epsilon, potential.sigma, potential.cutoff, potential.shift == "auto")


class _VerletListTabulated(_VerletListInteraction):
    label = "tab"

    def setPotential(self, type1, type2, potential):
        self._pots[(min(type1, type2), max(type1, type2))] = potential
        self.system.engine.nb_table(type1, type2, potential.r0, potential.dr, potential.e, potential.f, potential.cutoff)


class _VerletListMixedTabulated(_VerletListInteraction):
    """interaction.VerletListMixedTabulated(vl).setPotential(type1, type2, MixedTabulated(...)) (gromacs_topology.py:756-790)."""
    label = "mix_tab"

    def setPotential(self, type1, type2, potential):
        self._pots[(min(type1, type2), max(type1, type2))] = potential

        def push(x, t1=type1, t2=type2, pot=potential):
            x = min(max(float(x), 0.0), 1.0)
            if getattr(pot, "_sent", {}).get((t1, t2)) == x:
                return
            e, f = pot.rows(x)
            self.system.engine.nb_table(t1, t2, pot.r0, pot.dr, e, f, pot.cutoff)
            pot.__dict__.setdefault("_sent", {})[(t1, t2)] = x
            pot.mix_value = x
        if potential.observable is not None:
            potential.observable.connect(push)
            push(potential.observable.compute() if self.system.engine.n else 0.0)
        else:
            push(potential.mix_value)


class _FixedListInteraction(object):
    by_types = False

    def _kp(self, pot):
        """(kind, parameter list) of a potential object; a Tabulated bond potential (func 8,
        gromacs_topology.py:919-925) registers its rows once per engine and passes the table handle."""
        if isinstance(pot, _Tabulated):
            kind = "ANG_TABULATED" if isinstance(pot, _TabulatedAngular) else ("DIH_TABULATED" if isinstance(pot, _TabulatedDihedral) else "TABULATED")
            eng = self.system.engine
            cache = pot.__dict__.setdefault("_handles", {})
            if id(eng) not in cache:
                cache[id(eng)] = eng.table_create(pot.r0, pot.dr, pot.e, pot.f)
            return kind, [float(cache[id(eng)])]
        if isinstance(pot, _LennardJones):      # FixedPairList[Types]LennardJones: 1-4 pairs
            return "LJ_BOND", [pot.epsilon, pot.sigma, pot.cutoff if pot.cutoff is not None else 1e30]
        return pot.kind, pot.params()

    def __init__(self, system, flist, potential=None):
        self.system, self.flist = system, flist
        self.potential = potential
        self._typed = {}
        if potential is not None:
            kind, par = self._kp(potential)
            h = flist._bind(kind, False)
            system.engine.list_set_params(h, par)

    def setPotential(self, *args):
        pot = args[-1]
        types = tuple(int(t) for t in args[:-1])
        kind, par = self._kp(pot)
        if not self.by_types:
            self.potential = pot
            h = self.flist._bind(kind, False)
            self.system.engine.list_set_params(h, par)
            return
        h = self.flist._bind(kind, True)
        self._typed[types] = pot
        self.system.engine.list_set_params(h, par, types=types)

    def getFixedPairList(self):
        return self.flist

    getFixedTripleList = getFixedQuadrupleList = getFixedPairList


class _FixedListTypesInteraction(_FixedListInteraction):
    by_types = True

    def __init__(self, system, flist):
        self.system, self.flist = system, flist
        self.potential = None
        self._typed = {}


def _ns(**kw):
    return types.SimpleNamespace(**kw)


interaction = _ns(
    LennardJones=_LennardJones, Tabulated=_Tabulated, Harmonic=_Harmonic, FENE=_FENE,
    AngularHarmonic=_AngularHarmonic, Cosine=_Cosine, DihedralHarmonicNCos=_DihedralHarmonicNCos, DihedralRB=_DihedralRB,
    VerletListLennardJones=_VerletListLennardJones, VerletListTabulated=_VerletListTabulated,
    FixedPairListHarmonic=_FixedListInteraction, FixedPairListFENE=_FixedListInteraction,
    FixedPairListTypesHarmonic=_FixedListTypesInteraction, FixedPairListTypesFENE=_FixedListTypesInteraction,
    FixedTripleListAngularHarmonic=_FixedListInteraction, FixedTripleListCosine=_FixedListInteraction,
    FixedTripleListTypesAngularHarmonic=_FixedListTypesInteraction, FixedTripleListTypesCosine=_FixedListTypesInteraction,
    FixedQuadrupleListDihedralHarmonicNCos=_FixedListInteraction, FixedQuadrupleListDihedralRB=_FixedListInteraction,
    FixedQuadrupleListTypesDihedralHarmonicNCos=_FixedListTypesInteraction, FixedQuadrupleListTypesDihedralRB=_FixedListTypesInteraction,
    FENELennardJones=_FENELennardJones, DihedralHarmonic=_DihedralHarmonic,
    FixedPairListFENELennardJones=_FixedListInteraction, FixedPairListTypesFENELennardJones=_FixedListTypesInteraction,
    FixedPairListLennardJones=_FixedListInteraction, FixedPairListTypesLennardJones=_FixedListTypesInteraction,
    FixedQuadrupleListDihedralHarmonic=_FixedListInteraction, FixedQuadrupleListTypesDihedralHarmonic=_FixedListTypesInteraction,
    # out of scope
    CoulombTruncated=_unsupported("interaction.CoulombTruncated"),
    VerletListCoulombTruncated=_unsupported("interaction.VerletListCoulombTruncated"),
    TabulatedAngular=_TabulatedAngular, TabulatedDihedral=_TabulatedDihedral,
    FixedQuadrupleListTabulatedDihedral=_FixedListInteraction, FixedQuadrupleListTypesTabulatedDihedral=_FixedListTypesInteraction,
    FixedPairListTabulated=_FixedListInteraction, FixedPairListTypesTabulated=_FixedListTypesInteraction,
    FixedTripleListTabulatedAngular=_FixedListInteraction, FixedTripleListTypesTabulatedAngular=_FixedListTypesInteraction,
    FixedPairListLambdaHarmonic=_unsupported("interaction.FixedPairListLambdaHarmonic"),
    VerletListDynamicResolutionLennardJones=_unsupported("interaction.VerletListDynamicResolutionLennardJones"),
    MixedTabulated=_MixedTabulated, VerletListMixedTabulated=_VerletListMixedTabulated, MultiTabulated=_unsupported("interaction.MultiTabulated"),
)


# ---- integrator + extensions -------------------------------------------------------------------
class _VelocityVerlet(object):
    """integrator.VelocityVerlet(system): .dt .step run(n) addExtension getTimers
    (start_simulation.py:165-167,780)."""

    def __init__(self, system):
        self.system = system
        system.integrator = self
        self._dt = None
        self._ext = []

    @property
    def dt(self):
        return self._dt

    @dt.setter
    def dt(self, value):
        self._dt = float(value)
        self.system.engine.set_dt(value)

    @property
    def step(self):
        return self.system.engine.step

    def addExtension(self, ext):
        self._ext.append(ext)
        if hasattr(ext, "_connect"):
            ext._connect(self)

    def getNumberOfExtensions(self):
        return len(self._ext)

    def getExtension(self, k):
        return self._ext[k]

    def run(self, nsteps):
        # ExtAnalyze observers fire every `interval` steps (aftIntV): split the call at those boundaries
        e = self.system.engine
        if self.system.storage is not None:
            self.system.storage.decompose()
        ana = [x for x in self._ext if isinstance(x, _ExtAnalyze)]
        done = 0
        while done < nsteps:
            chunk = nsteps - done
            for a in ana:
                chunk = min(chunk, a.interval - (e.step % a.interval))
            e.run(chunk)
            done += chunk
            for a in ana:
                if e.step % a.interval == 0:
                    a.perform()
        if nsteps == 0:
            e.run(0)

    def getTimers(self):
        t = self.system.engine.timers()
        return [("Run", t["run_wall_s"]), ("Reaction", t["reaction_wall_s"]), ("Resort", t["rebuild_wall_s"])]


class _LangevinThermostat(object):
    """integrator.LangevinThermostat(system): .temperature (= T*kb), .gamma (start_simulation.py:330-336)."""

    def __init__(self, system):
        self.system = system
        self.temperature = 0.0
        self.gamma = 0.0
        self.valid_types = None

    def add_valid_types(self, types_):
        """Thermal groups (start_simulation.py:312-336): only particles whose current type is listed are thermalised;
        an empty list (no --thermal_groups / --table_groups) leaves every type thermalised."""
        self.valid_types = sorted(set(int(t) for t in types_ if t is not None))
        if getattr(self, "_connected", False):
            self.system.engine.thermostat_langevin_types(self.valid_types)

    def _connect(self, integrator):
        seed = self.system.rng.get_seed() if self.system.rng is not None else 0
        self.system.engine.thermostat_langevin(self.temperature, self.gamma, seed)
        self.system.engine.thermostat_langevin_types(self.valid_types or [])
        self._connected = True


class _BerendsenThermostat(object):
    """integrator.BerendsenThermostat(system): .temperature (= T*kb), .tau (start_simulation.py:341-344)."""

    def __init__(self, system):
        self.system, self.temperature, self.tau = system, 0.0, 1.0

    def _connect(self, integrator):
        self.system.engine.thermostat_rescale("berendsen", self.temperature, self.tau)

    def disconnect(self):
        self.system.engine.thermostat_rescale(None, 1.0, 1.0)


class _Isokinetic(object):
    """integrator.Isokinetic(system): .temperature, .coupling = rescale every `coupling` steps (start_simulation.py:345-348)."""

    def __init__(self, system):
        self.system, self.temperature, self.coupling = system, 0.0, 1

    def _connect(self, integrator):
        self.system.engine.thermostat_rescale("isokinetic", self.temperature, int(self.coupling))

    def disconnect(self):
        self.system.engine.thermostat_rescale(None, 1.0, 1.0)


class _StochasticVelocityRescaling(object):
    """integrator.StochasticVelocityRescaling(system): .temperature (= T*kb), .coupling (start_simulation.py:337-340)."""

    def __init__(self, system):
        self.system, self.temperature, self.coupling = system, 0.0, 1.0

    def _connect(self, integrator):
        seed = self.system.rng.get_seed() if self.system.rng is not None else 0
        self.system.engine.thermostat_svr(self.temperature, self.coupling, seed)

    def disconnect(self):
        self.system.engine.thermostat_svr(1.0, 0.0, 0)


class _CapForce(object):
    """integrator.CapForce(system, max_force) (start_simulation.py:320-324): conservative force of a particle
    rescaled to |f| = max_force where it exceeds it, before the thermostat's terms."""

    def __init__(self, system, capForce, particleGroup=None):
        if particleGroup is not None:
            raise NotImplementedError("CapForce on a particle group is outside the hot-path scope")
        if not isinstance(capForce, (int, float)):
            raise NotImplementedError("CapForce with a Real3D cap is outside the hot-path scope (ChemLab passes a scalar)")
        self.system = system
        self.capForce = float(capForce)

    def _connect(self, integrator):
        self.system.engine.cap_force(self.capForce)

    def disconnect(self):
        self.system.engine.cap_force(0.0)


class _TopologyParticleProperties(object):
    """integrator.TopologyParticleProperties(type=, mass=, q=, state=, incr_state=) + set_min_max_state(min, max)
    (reaction_setup.py:245-249: the neighbour changes only while its state is in [min, max), and its state is incremented)."""

    def __init__(self, type=None, mass=None, q=None, state=None, incr_state=None, **kw):
        self.type, self.mass, self.q, self.state, self.incr_state = type, mass, q, state, incr_state
        self.min_state = self.max_state = None

    def set_min_max_state(self, min_state, max_state):
        self.min_state, self.max_state = int(min_state), int(max_state)


class _ReactionConstraintNeighbourState(object):
    """integrator.ReactionConstraintNeighbourState(type_id, min_state, max_state): the constrained reactant must have a
    bonded neighbour of that type in that state window (reaction_setup.py:203-204)."""

    def __init__(self, type_id, min_state, max_state):
        self.type_id, self.min_state, self.max_state = int(type_id), int(min_state), int(max_state)


class _ATRPActivator(object):
    """integrator.ATRPActivator(system, interval, num_particles, ratio_activator, ratio_deactivator, delta_catalyst,
    k_activate, k_deactivate): .stats_filename, .select_from_all, add_reactive_center(type_id, state, is_activator,
    new_property, delta_state) (reaction_post_process.py:380-426).  Bound to chem_atrp_* when added to the integrator."""

    def __init__(self, system, interval, num_particles, ratio_activator, ratio_deactivator, delta_catalyst, k_activate, k_deactivate):
        self.system = system
        self.interval, self.num_particles = int(interval), int(num_particles)
        self.ratio_activator, self.ratio_deactivator = float(ratio_activator), float(ratio_deactivator)
        self.delta_catalyst, self.k_activate, self.k_deactivate = float(delta_catalyst), float(k_activate), float(k_deactivate)
        self.stats_filename = None
        self.select_from_all = 1
        self._centers = []
        self._connected = False

    def add_reactive_center(self, type_id, state, is_activator, new_property, delta_state):
        c = (int(type_id), int(state), bool(is_activator), int(new_property.type), float(new_property.mass), float(new_property.q or 0.0), int(delta_state))
        self._centers.append(c)
        if self._connected:
            self.system.engine.atrp_add_center(*c)

    def _connect(self, integrator):
        e = self.system.engine
        seed = self.system.rng.get_seed() if self.system.rng is not None else 0
        if not self._connected:
            for c in self._centers:
                e.atrp_add_center(*c)
        e.atrp_init(self.interval, self.num_particles, self.ratio_activator, self.ratio_deactivator, self.delta_catalyst,
                    self.k_activate, self.k_deactivate, select_from_all=bool(int(self.select_from_all)), seed=seed)
        self._connected = True

    def disconnect(self):
        self.system.engine.atrp_disconnect()

    def save_stats(self, filename=None):
        """One row per firing: step, activator and deactivator fractions, flips (the reference's `stats_file`)."""
        filename = filename or self.stats_filename
        rows = self.system.engine.atrp_stats()
        if filename:
            with _rank.wopen(filename, "w") as f:
                f.write("# step ratio_activator ratio_deactivator activated deactivated candidates selected\n")
                for r in rows:
                    f.write("%d %.10g %.10g %d %d %d %d\n" % (r["step"], r["ratio_activator"], r["ratio_deactivator"], r["activated"],
                                                              r["deactivated"], r["candidates"], r["selected"]))
        return rows


class _PostProcessChangeNeighboursProperty(object):
    """integrator.PostProcessChangeNeighboursProperty(topology_manager).add_change_property(old_type, props, nb_level)
    (reaction_post_process.py:76-115): attached to a reaction with add_postprocess(pp, 'type_1' | 'type_2' | 'both')."""

    def __init__(self, topology_manager=None):
        self.rules = []     # (old_type, TopologyParticleProperties, nb_level) in insertion order

    def add_change_property(self, type_id, prop, nb_level):
        self.rules.append((int(type_id), prop, int(nb_level)))


class _PostProcessChangeProperty(object):
    """integrator.PostProcessChangeProperty() and PostProcessChangePropertyByTopologyManager(tm) (reaction_setup.py:225-236):
    both change type / mass / charge of the reactant itself; here they are the same thing (new_type_* of chem_reaction_desc)."""

    def __init__(self, topology_manager=None):
        self.changes = {}

    def add_change_property(self, type_id, prop):
        self.changes[int(type_id)] = prop


class _ReactionCutoff(object):
    def __init__(self, cutoff):
        self.cutoff = cutoff
        self.min_cutoff = 0.0


class _Reaction(object):
    """integrator.Reaction(type_1, type_2, delta_1, delta_2, min_state_1, max_state_1, min_state_2,
    max_state_2, rate, fpl, cutoff) (reaction_setup.py:81-92)."""

    def __init__(self, type_1, type_2, delta_1, delta_2, min_state_1, max_state_1, min_state_2, max_state_2, rate, fpl, cutoff):
        self.__dict__.update(type_1=type_1, type_2=type_2, delta_1=delta_1, delta_2=delta_2, min_state_1=min_state_1,
                             max_state_1=max_state_1, min_state_2=min_state_2, max_state_2=max_state_2, rate=rate, fpl=fpl)
        self._cut = _ReactionCutoff(cutoff)
        self.intramolecular = False
        self.intraresidual = False
        self.is_virtual = False
        self.active = True
        self._pp = {}
        self._nb_pp = []
        self._constraints = []
        self._index = None
        self._system = None

    def add_constraint(self, constraint, which="type_1"):
        if not isinstance(constraint, _ReactionConstraintNeighbourState):
            raise NotImplementedError("reaction constraint %s is outside the hot-path scope" % type(constraint).__name__)
        self._constraints.append((constraint, which))

    @property
    def cutoff(self):
        return self._cut.cutoff

    def get_reaction_cutoff(self):
        return self._cut

    def set_reaction_cutoff(self, rc):
        raise NotImplementedError("ReactionCutoffRandom is outside the hot-path scope")

    def add_postprocess(self, pp, which="type_1"):
        if isinstance(pp, _PostProcessChangeNeighboursProperty):
            self._nb_pp.append((pp, which))
            return
        if not isinstance(pp, _PostProcessChangeProperty):
            raise NotImplementedError("post-process %s is outside the hot-path scope" % type(pp).__name__)
        self._pp[which] = pp

    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)
        if k == "rate" and getattr(self, "_index", None) is not None:
            self._system.engine.reaction_set_rate(self._index, v)


class _RestrictReaction(_Reaction):
    """integrator.RestrictReaction(...): a Reaction that only accepts the id pairs given by define_connection(b1, b2)
    (group option `connectivity_map`, reaction_setup.py:75-78,115-128)."""

    def __init__(self, *a, **kw):
        _Reaction.__init__(self, *a, **kw)
        object.__setattr__(self, "_connections", [])
        object.__setattr__(self, "revert", False)

    def define_connection(self, b1, b2):
        self._connections.append((int(b1), int(b2)))
        if self._index is not None:
            self._system.engine.reaction_restrict(self._index, [(int(b1), int(b2))])


class _ChemicalReaction(object):
    """integrator.ChemicalReaction(system, vl, storage, topology_manager, interval): nearest_mode,
    max_per_interval, add_reaction, disconnect (reaction_setup.py:416-427,506)."""

    def __init__(self, system, vl, storage, topology_manager, interval):
        self.system, self.interval = system, int(interval)
        self.nearest_mode = False
        self.max_per_interval = -1
        self._reactions = []
        self._initialised = False

    def add_reaction(self, r):
        self._reactions.append(r)

    def _flush(self):
        e = self.system.engine
        if not self._initialised:
            seed = self.system.rng.get_seed() if self.system.rng is not None else 0
            e.reaction_init(self.interval, bool(self.nearest_mode), max(int(self.max_per_interval), 0), seed)
            self._initialised = True
        for r in self._reactions:
            if r._index is not None:
                continue
            kw = {}
            for which, k in (("type_1", 1), ("type_2", 2)):
                pp = r._pp.get(which)
                old = getattr(r, which)
                if pp is not None and old in pp.changes:
                    ch = pp.changes[old]
                    kw["new_type_%d" % k] = int(ch.type)
                    kw["new_mass_%d" % k] = float(ch.mass)
                    kw["new_q_%d" % k] = float(ch.q or 0.0)
            bond_list = -1
            if not r.is_virtual:
                if r.fpl.handle is None:
                    raise RuntimeError("Reaction: the FixedPairList has no interaction (potential) attached")
                bond_list = r.fpl.handle
            r._index = e.reaction_add(r.type_1, r.type_2, r.delta_1, r.delta_2, r.min_state_1, r.max_state_1, r.min_state_2,
                                      r.max_state_2, r.rate, r.cutoff, bond_list=bond_list, min_cutoff=r._cut.min_cutoff,
                                      intramolecular=r.intramolecular, intraresidual=r.intraresidual, is_virtual=r.is_virtual,
                                      active=r.active, **kw)
            r._system = self.system
            if isinstance(r, _RestrictReaction):
                if r.revert:
                    raise NotImplementedError("RestrictReaction.revert (dissociation along a connectivity map) is outside the hot-path scope")
                e.reaction_restrict(r._index, r._connections)
            for cons, which in r._constraints:
                e.reaction_constraint(r._index, which, cons.type_id, cons.min_state, cons.max_state)
            for pp, which in r._nb_pp:
                for old_type, prop, nb_level in pp.rules:
                    window = None if getattr(prop, "min_state", None) is None else (prop.min_state, prop.max_state)
                    e.reaction_neighbour_change(r._index, which, old_type, nb_level, int(prop.type), float(prop.mass),
                                                float(prop.q or 0.0), None if prop.state is None else int(prop.state),
                                                incr_state=getattr(prop, "incr_state", None), state_window=window)

    def _connect(self, integrator):
        self._flush()
        self.system.engine.reactions_enable(True)

    def disconnect(self):
        self.system.engine.reactions_enable(False)

    def connect(self):
        self.system.engine.reactions_enable(True)

    def get_timers(self):
        return []

    def save_reaction_counters(self, filename):
        """events per reaction index (start_simulation.py:1028)"""
        ev = self.system.engine.get_events()
        with _rank.wopen(filename, "w") as f:
            for r in range(len(self._reactions)):
                f.write("%d %d\n" % (r, int((ev["reaction"] == r).sum())))

    def save_intra_inter_counter(self, filename):
        """events between particles of the same / of different bonded clusters at the time of the event, per reaction
        (start_simulation.py:1034); needs engine option count_intra_inter (the driver sets it)."""
        ev = self.system.engine.get_events()
        with _rank.wopen(filename, "w") as f:
            f.write("# reaction intra inter\n")
            for r in range(len(self._reactions)):
                m = ev["reaction"] == r
                f.write("%d %d %d\n" % (r, int((ev["pad"][m] == 1).sum()), int((ev["pad"][m] == 0).sum())))


class _TopologyManager(object):
    """integrator.TopologyManager(system): observe_tuple, register_tuple/triplet/quadruplet,
    initialize_topology (start_simulation.py:211-212,395-440).  The bond graph itself lives in the
    library (chem_host.hpp); this object only forwards the registrations."""

    def __init__(self, system):
        self.system = system
        self._fpls = []

    def observe_tuple(self, fpl):
        if getattr(fpl, "arity", 2) == 2 and fpl not in self._fpls:
            self._fpls.append(fpl)

    def observe_triple(self, ftl):
        pass

    observe_quadruple = observe_triple

    # end-of-run dumps (start_simulation.py:1004-1006); layout: chemlab/outputs.py write_topology_dumps
    def _dump(self, filename, suffix):
        from ..chemlab import outputs
        import os, shutil, tempfile
        tmp = tempfile.mkdtemp()
        try:
            outputs.write_topology_dumps(os.path.join(tmp, "x"), self.system, self._fpls)
            if not _rank.is_root():      # (the dump above gathers collectively: every rank makes it, rank 0 keeps the file)
                return
            d = os.path.dirname(filename)
            if d and not os.path.isdir(d):
                os.makedirs(d)
            shutil.move(os.path.join(tmp, "x" + suffix), filename)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    def save_topology(self, filename):
        self._dump(filename, "_topology.dat")

    def save_res_topology(self, filename):
        self._dump(filename, "_res_topology.dat")

    def save_residues(self, filename):
        self._dump(filename, "_residue_list.dat")

    def register_tuple(self, fpl, t1, t2):
        pass   # bonds never spawn from other bonds

    def register_triplet(self, ftl, t1, t2, t3):
        if ftl.handle is None:
            raise RuntimeError("register_triplet: the FixedTripleList has no interaction attached")
        self.system.engine.topology_register(ftl.handle, [t1, t2, t3])

    def register_quadruplet(self, fql, t1, t2, t3, t4):
        if fql.handle is None:
            raise RuntimeError("register_quadruplet: the FixedQuadrupleList has no interaction attached")
        self.system.engine.topology_register(fql.handle, [t1, t2, t3, t4])

    def initialize_topology(self):
        pass

    def _connect(self, integrator):
        pass

    def get_timers(self):
        return []


class _ExtAnalyze(object):
    def __init__(self, obj, interval):
        self.obj, self.interval = obj, max(1, int(interval))

    def perform(self):
        if hasattr(self.obj, "perform_action"):
            self.obj.perform_action()
        elif hasattr(self.obj, "perform"):
            self.obj.perform()

    def _connect(self, integrator):
        pass


integrator = _ns(
    VelocityVerlet=_VelocityVerlet, LangevinThermostat=_LangevinThermostat, ChemicalReaction=_ChemicalReaction,
    Reaction=_Reaction, PostProcessChangeProperty=_PostProcessChangeProperty,
    PostProcessChangePropertyByTopologyManager=_PostProcessChangeProperty, ReactionConstraintNeighbourState=_ReactionConstraintNeighbourState,
    TopologyParticleProperties=_TopologyParticleProperties, TopologyManager=_TopologyManager, ExtAnalyze=_ExtAnalyze,
    StochasticVelocityRescaling=_StochasticVelocityRescaling,
    BerendsenThermostat=_BerendsenThermostat, BerendsenBarostat=_unsupported("integrator.BerendsenBarostat"),
    Isokinetic=_Isokinetic, LangevinBarostat=_unsupported("integrator.LangevinBarostat"),
    CapForce=_CapForce, RestrictReaction=_RestrictReaction,
    DissociationReaction=_unsupported("integrator.DissociationReaction"), ATRPActivator=_ATRPActivator,
    ReactionCutoffRandom=_unsupported("integrator.ReactionCutoffRandom"), FixDistances=_unsupported("integrator.FixDistances"),
    ChangeInRegion=_unsupported("integrator.ChangeInRegion"), BasicDynamicResolution=_unsupported("integrator.BasicDynamicResolution"),
    PostProcessChangeNeighboursProperty=_PostProcessChangeNeighboursProperty,
    PostProcessRemoveNeighbourBond=_unsupported("integrator.PostProcessRemoveNeighbourBond"),
    PostProcessJoinParticles=_unsupported("integrator.PostProcessJoinParticles"),
)


# ---- analysis ------------------------------------------------------------------------------------
class _Observable(object):
    def __init__(self, system, *a):
        self.system, self.args = system, a

    def compute(self):
        raise NotImplementedError


class _Temperature(_Observable):
    def compute(self):
        return self.system.engine.observe()["temperature"]


class _KineticEnergy(_Observable):
    def __init__(self, system, temperature=None):
        _Observable.__init__(self, system)

    def compute(self):
        return self.system.engine.observe()["ekin"]


class _PotentialEnergy(_Observable):
    def __init__(self, system, interaction_, compute_method=None):
        _Observable.__init__(self, system)
        self.interaction = interaction_

    def compute(self):
        o = self.system.engine.observe()
        i = self.interaction
        if isinstance(i, _VerletListLennardJones):
            return o["epot_lj"]
        if isinstance(i, (_VerletListTabulated, _VerletListMixedTabulated)):    # (one tabulated-energy accumulator: plain and mixed tables together)
            return o["epot_tab"]
        h = i.flist.handle
        return o["epot_list"][h] if h is not None else 0.0


class _NPart(_Observable):
    def compute(self):
        return self.system.engine.n


class _MaxPID(_Observable):
    def compute(self):
        return int(self.system.engine.get_state("ID").max())


class _NFixedPairListEntries(_Observable):
    def __init__(self, system, fpl):
        _Observable.__init__(self, system)
        self.fpl = fpl

    def compute(self):
        return self.fpl.totalSize()


class _CMVelocity(_Observable):
    def reset(self):
        pass   # the synthetic/initial velocities are generated with zero COM momentum

    def compute(self):
        o = self.system.engine.observe()
        return Real3D(*o["momentum"])


class _ChemicalConversion(_Observable):
    def __init__(self, system, type_id, total=None):
        _Observable.__init__(self, system)
        self.type_id, self.total = type_id, total

    def compute(self):
        t = self.system.engine.get_state("TYPE")
        c = float((t == self.type_id).sum())
        val = c / self.total if self.total else c
        for fn in getattr(self, "_listeners", ()):       # the reference's onValue signal: MixedTabulated follows the conversion
            fn(val)
        return val

    def connect(self, fn):
        self.__dict__.setdefault("_listeners", []).append(fn)


class _ChemicalConversionTypeState(_Observable):
    def __init__(self, system, type_id, state, total=None):
        _Observable.__init__(self, system)
        self.type_id, self.state, self.total = type_id, state, total

    def compute(self):
        e = self.system.engine
        c = float(((e.get_state("TYPE") == self.type_id) & (e.get_state("STATE") == self.state)).sum())
        return c / self.total if self.total else c


class _SystemMonitorOutputCSV(object):
    def __init__(self, filename, delimiter=","):
        self.filename, self.delimiter = filename, delimiter
        self._header_written = False

    def write(self, names, row):
        with _rank.wopen(self.filename, "a") as f:
            if not self._header_written:
                f.write(self.delimiter.join(names) + "\n")
                self._header_written = True
            f.write(self.delimiter.join("%g" % v for v in row) + "\n")


class _SystemMonitor(object):
    """analysis.SystemMonitor(system, integrator, output) + add_observable/info
    (start_simulation.py:447-569)."""

    def __init__(self, system, integrator_, output):
        self.system, self.integrator, self.output = system, integrator_, output
        self.obs = []
        self.last = None
        self.copy_state = None

    def add_observable(self, name, observable, visible=True):
        self.obs.append((name, observable, visible))

    def perform_action(self):
        names = ["step", "time"] + [n for n, _, _ in self.obs]
        row = [self.integrator.step, self.integrator.step * (self.integrator.dt or 0.0)] + [float(o.compute()) for _, o, _ in self.obs]
        self.last = (names, row)
        if self.output is not None:
            self.output.write(names, row)

    def info(self):
        if self.last is None:
            return
        names, row = self.last
        vis = ["step", "time"] + [n for n, _, v in self.obs if v]
        print(" ".join("%s=%g" % (n, v) for n, v in zip(names, row) if n in vis))

    def dump(self):
        pass


analysis = _ns(
    Temperature=_Temperature, KineticEnergy=_KineticEnergy, PotentialEnergy=_PotentialEnergy, NPart=_NPart, MaxPID=_MaxPID,
    NFixedPairListEntries=_NFixedPairListEntries, CMVelocity=_CMVelocity, ChemicalConversion=_ChemicalConversion,
    ChemicalConversionTypeState=_ChemicalConversionTypeState, SystemMonitor=_SystemMonitor,
    SystemMonitorOutputCSV=_SystemMonitorOutputCSV,
    Pressure=_unsupported("analysis.Pressure"), PressureTensor=_unsupported("analysis.PressureTensor"),
)

esutil = _ns(RNG=_RNG)
bc = _ns(OrthorhombicBC=_OrthorhombicBC)
storage = _ns(DomainDecomposition=_DomainDecomposition)


def _gaussian_velocities(T, n, masses, kb=1.0, seed=0):
    """tools.velocities.gaussian: Maxwell velocities with zero COM momentum (start_simulation.py:136-146)."""
    rng = np.random.default_rng(seed)
    m = np.asarray(masses, dtype=np.float64)
    v = rng.standard_normal((n, 3)) * np.sqrt(kb * T / m)[:, None]
    v -= (m[:, None] * v).sum(0) / m.sum()
    return v[:, 0], v[:, 1], v[:, 2]


tools = _ns(decomp=_ns(nodeGrid=_node_grid, cellGrid=_cell_grid, tuneSkin=_unsupported("tools.decomp.tuneSkin")),
            velocities=_ns(gaussian=_gaussian_velocities),
            analyse=_ns(final_info=lambda *a, **k: None))
class _DumpGRO(object):
    """io.DumpGRO(system, integrator, filename, unfolded=False, append=False): every particle, generic names
    (start_simulation.py:1014-1016)."""

    def __init__(self, system, integrator, filename="out.gro", unfolded=False, append=False, **kw):
        self.system, self.filename, self.unfolded, self.append = system, filename, unfolded, append

    def dump(self):
        e = self.system.engine
        ids, pos = e.get_state("ID"), e.get_state("POS_UNFOLDED" if self.unfolded else "POS")
        vel, types, res = e.get_state("VEL"), e.get_state("TYPE"), e.get_state("RESID")
        box = list(self.system.bc.boxL)
        with _rank.wopen(self.filename, "a" if self.append else "w") as f:
            f.write("system size %d\n%d\n" % (len(ids), len(ids)))
            for k in range(len(ids)):
                f.write("%5d%-5s%5s%5d%8.3f%8.3f%8.3f%8.4f%8.4f%8.4f\n" % (int(res[k]) % 100000, "T%d" % types[k], "T%d" % types[k], int(ids[k]) % 100000,
                                                                            pos[k, 0], pos[k, 1], pos[k, 2], vel[k, 0], vel[k, 1], vel[k, 2]))
            f.write("%f %f %f\n" % tuple(box))

    perform_action = dump


class _DumpH5MD(object):
    """io.DumpH5MD(system, filename, group_name='atoms', store_species=..., store_state=..., ...) -- the trajectory writer
    of start_simulation.py:571-589: dump(step, time) appends one frame, flush() writes the file, close().
    Layout = the H5MD tree analysis scripts read (examples/chain_growth_catalytic/Checkup.ipynb,
    examples/mf/espp_cg_1_water/analyze.py:20-24):
        /particles/<group>/{position,image,species,state,res_id,mass,id,velocity,force}/{value,step,time}   value: [frames, N(, 3)]
        /particles/<group>/box/edges/{value,step,time}                                                   value: [frames, 3]
        /connectivity/<name>/{value,step,time}     value: [frames, max entries, arity], unused rows filled with -1
    With h5py importable the file is real HDF5; otherwise (this image) the same tree is written as a NumPy .npz archive
    whose keys are the dataset paths (`<filename minus .h5>.npz`); tools/npz2h5md.py turns it into the .h5 file."""

    def __init__(self, system, filename, group_name="atoms", static_box=True, author="", email="", store_species=True, store_res_id=True,
                 store_charge=False, store_position=True, store_state=True, store_lambda=False, store_force=False, store_velocity=False,
                 store_mass=True, is_single_prec=True, chunk_size=256, **kw):
        self.system, self.filename, self.group = system, filename, group_name
        self.store = dict(species=store_species, res_id=store_res_id, position=store_position, state=store_state, force=store_force,
                          velocity=store_velocity, mass=store_mass)
        self.real = np.float32 if is_single_prec else np.float64
        self.frames = {}            # dataset path -> list of per-frame arrays
        self.steps, self.times = [], []
        self.connectivity = {}      # name -> dict(value=[...], step=[...], time=[...], arity)
        self.static_connectivity = {}

    def _add(self, name, arr):
        self.frames.setdefault("particles/%s/%s/value" % (self.group, name), []).append(arr)

    def dump(self, step, time_):
        e = self.system.engine
        self.steps.append(int(step)); self.times.append(float(time_))
        self._add("id", e.get_state("ID").astype(np.int64))
        if self.store["position"]:
            self._add("position", e.get_state("POS").astype(self.real))
            self._add("image", e.get_state("IMAGE").astype(np.int32))
        if self.store["velocity"]:
            self._add("velocity", e.get_state("VEL").astype(self.real))
        if self.store["force"]:
            self._add("force", e.get_state("FORCE").astype(self.real))
        if self.store["species"]:
            self._add("species", e.get_state("TYPE").astype(np.int32))
        if self.store["state"]:
            self._add("state", e.get_state("STATE").astype(np.int32))
        if self.store["res_id"]:
            self._add("res_id", e.get_state("RESID").astype(np.int32))
        if self.store["mass"]:
            self._add("mass", e.get_state("MASS").astype(self.real))
        self.frames.setdefault("particles/%s/box/edges/value" % self.group, []).append(np.asarray(list(self.system.bc.boxL), dtype=self.real))

    def tree(self):
        """dataset path -> array, as written by flush()."""
        out = {}
        st, tm = np.asarray(self.steps, dtype=np.int64), np.asarray(self.times, dtype=np.float64)
        for path, fr in self.frames.items():
            out[path] = np.stack(fr) if fr else np.zeros((0,))
            base = path[:-len("/value")]
            out[base + "/step"], out[base + "/time"] = st[:len(fr)], tm[:len(fr)]
        for name, c in self.connectivity.items():
            rows = max([len(v) for v in c["value"]] + [1])
            val = -np.ones((len(c["value"]), rows, c["arity"]), dtype=np.int64)
            for k, v in enumerate(c["value"]):
                if len(v):
                    val[k, :len(v)] = v
            out["connectivity/%s/value" % name] = val
            out["connectivity/%s/step" % name] = np.asarray(c["step"], dtype=np.int64)
            out["connectivity/%s/time" % name] = np.asarray(c["time"], dtype=np.float64)
        for name, v in self.static_connectivity.items():
            out["connectivity/%s" % name] = v
        return out

    def flush(self):
        tree = self.tree()
        if not _rank.is_root():
            return self.filename
        try:
            import h5py
        except ImportError:
            h5py = None
        if h5py is not None:
            with h5py.File(self.filename, "w") as h5:
                for path, arr in tree.items():
                    h5.create_dataset(path, data=arr)
            return self.filename
        out = self.filename[:-3] + ".npz" if self.filename.endswith(".h5") else self.filename + ".npz"
        np.savez_compressed(out, **tree)
        return out

    def close(self):
        return self.flush()


class _DumpTopology(object):
    """io.DumpTopology(system, integrator, h5md_file): observe_tuple/triple/quadruple(list, name) record the CURRENT entries of
    a (growing) list at every dump() into /connectivity/<name> (-1 padded rows); add_static_* store a list once
    (start_simulation.py:591-642).  Used through ExtAnalyze(dump_topol, topol_collect); update() is the reference's
    flush-to-file hook (a no-op here: the trajectory object writes everything in flush())."""

    def __init__(self, system, integrator, h5md_file):
        self.system, self.integrator, self.h5 = system, integrator, h5md_file
        self._observed = []

    def _observe(self, lst, name, arity):
        self._observed.append((lst, name))
        self.h5.connectivity[name] = dict(value=[], step=[], time=[], arity=arity)

    def observe_tuple(self, fpl, name):
        self._observe(fpl, name, 2)

    def observe_triple(self, ftl, name):
        self._observe(ftl, name, 3)

    def observe_quadruple(self, fql, name):
        self._observe(fql, name, 4)

    @staticmethod
    def _entries(lst):
        for getter in ("getAllBonds", "getAllTriples", "getAllQuadruples"):
            if hasattr(lst, getter):
                try:
                    return np.asarray(getattr(lst, getter)(), dtype=np.int64)
                except Exception:
                    continue
        return np.zeros((0, 2), dtype=np.int64)

    def add_static_tuple(self, fpl, name):
        self.h5.static_connectivity[name] = self._entries(fpl).reshape(-1, 2)

    def add_static_triple(self, ftl, name):
        self.h5.static_connectivity[name] = self._entries(ftl).reshape(-1, 3)

    def add_static_quadruple(self, fql, name):
        self.h5.static_connectivity[name] = self._entries(fql).reshape(-1, 4)

    def dump(self):
        step = self.integrator.step
        for lst, name in self._observed:
            c = self.h5.connectivity[name]
            c["value"].append(self._entries(lst).reshape(-1, c["arity"]))
            c["step"].append(int(step)); c["time"].append(float(step) * float(self.integrator.dt or 0.0))

    perform_action = dump
    perform = dump

    def update(self):
        pass


io = _ns(DumpH5MD=_DumpH5MD, DumpTopology=_DumpTopology, DumpGRO=_DumpGRO)
