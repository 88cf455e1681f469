"""Readers for the GROMACS-style inputs ChemLab consumes.

File formats and field meaning follow /root/reference/src/chemlab/files_io.py (GROFile.read
:161-214, GROMACSTopologyFile.read/_parse_* :497-533,613-820) and the `#include`/`#define`
pre-processing of gromacs_topology.py:60-107.  Only reading is in scope (writers: SURVEY f-3)."""
import collections
import os
import re

import numpy as np

Atom = collections.namedtuple("Atom", "atom_id name chain_name chain_idx position velocity")
TopAtom = collections.namedtuple("TopAtom", "atom_id atom_type chain_idx chain_name name cgnr charge mass molecule_name")


class GROFile(object):
    """Fixed-column .gro reader: atoms keyed by atom id, box from the last line."""
    scale_factor = 1.0

    def __init__(self, file_name):
        self.file_name = file_name
        self.atoms = {}
        self.box = None
        self.title = None

    def read(self):
        with open(self.file_name) as f:
            content = f.readlines()
        self.title = content[0].rstrip("\r\n")
        n = int(content[1])
        for line in content[2:n + 2]:
            chain_idx = int(line[0:5])
            chain_name = line[5:10].strip()
            at_name = line[10:15].strip()
            at_id = int(line[15:20])
            pos = np.array([float(line[20:28]), float(line[28:36]), float(line[36:44])]) * self.scale_factor
            vel = None
            if len(line.rstrip("\n")) > 45:
                vel = np.array([float(line[44:52]), float(line[52:60]), float(line[60:68])]) * self.scale_factor
            self.atoms[at_id] = Atom(at_id, at_name, chain_name, chain_idx, pos, vel)
        self.box = np.array([float(x) for x in content[n + 2].split()][:3]) * self.scale_factor
        return self


def preprocess_topology(file_name, defines=None):
    """Resolve `#include "file"` (relative to the including file), `#define NAME` and
    `#ifdef/#ifndef/#else/#endif`; returns the list of lines."""
    defines = set() if defines is None else defines
    out = []
    base = os.path.dirname(os.path.abspath(file_name))
    stack = []   # True = currently emitting
    with open(file_name) as f:
        for raw in f:
            line = raw.strip()
            if line.startswith("#ifdef") or line.startswith("#ifndef"):
                name = line.split()[1]
                cond = (name in defines) if line.startswith("#ifdef") else (name not in defines)
                stack.append(cond)
                continue
            if line.startswith("#else"):
                stack[-1] = not stack[-1]
                continue
            if line.startswith("#endif"):
                stack.pop()
                continue
            if stack and not all(stack):
                continue
            if line.startswith("#define"):
                defines.add(line.split()[1])
                continue
            if line.startswith("#include"):
                inc = re.search(r'#include\s+["<](.+?)[">]', line).group(1)
                out.extend(preprocess_topology(os.path.join(base, inc), defines))
                continue
            out.append(raw)
    return out


class GROMACSTopologyFile(object):
    """Section-wise .top reader with ChemLab's `[ atomstate ]` extension."""

    def __init__(self, file_name):
        self.file_name = file_name
        self.defaults = None
        self.atomtypes = {}
        self.atom_name2atomnr = {}
        self.atomnr2atom_name = collections.defaultdict(list)
        self.atomstate = {}
        self.nonbond_params = {}
        self.bondtypes = {}
        self.angletypes = {}
        self.dihedraltypes = {}
        self.moleculetype = collections.OrderedDict()
        self.molecules = []
        self.system_name = None
        self.current_molecule = None
        self.molecules_data = collections.defaultdict(dict)

    # -- sections --
    def _defaults(self, d):
        self.defaults = {"func": int(d[0]), "combinationrule": int(d[1]), "nbfunc": 1,
                         "gen-pairs": len(d) > 2 and d[2] == "yes",
                         "fudgeLJ": float(d[3]) if len(d) > 3 else 1.0, "fudgeQQ": float(d[4]) if len(d) > 4 else 1.0}

    def _atomtypes(self, d):
        if len(d) == 7:
            name, nr, mass, q, ptype, sig, eps = d[0], d[0], float(d[2]), float(d[3]), d[4], float(d[5]), float(d[6])
        elif len(d) == 6:
            name, nr, mass, q, ptype, sig, eps = d[0], d[0], float(d[1]), float(d[2]), d[3], float(d[4]), float(d[5])
        elif len(d) == 8 and d[0].startswith("opls"):
            name, nr, mass, q, ptype, sig, eps = d[0], d[1], float(d[3]), float(d[4]), d[5], float(d[6]), float(d[7])
        else:
            return
        self.atom_name2atomnr[name] = nr
        self.atomnr2atom_name[nr].append(name)
        self.atomtypes[name] = {"name": name, "mass": mass, "charge": q, "type": ptype, "sigma": sig, "epsilon": eps}
        if name in self.atomstate:
            self.atomtypes[name]["state"] = self.atomstate[name]

    def _atomstate(self, d):
        self.atomstate[d[0]] = int(d[1])
        if d[0] in self.atomtypes:
            self.atomtypes[d[0]]["state"] = int(d[1])

    def _nonbond_params(self, d):
        k = tuple(sorted(d[:2]))
        if k in self.nonbond_params:
            raise RuntimeError("%s already exists, wrong topology" % (k,))
        self.nonbond_params[k] = {"func": int(d[2]), "params": d[3:]}

    def _bondtypes(self, d):
        i, j = d[:2]
        v = {"func": int(d[2]), "params": d[3:]}
        self.bondtypes.setdefault(i, {})[j] = v
        self.bondtypes.setdefault(j, {})[i] = v

    def _angletypes(self, d):
        i, j, k = d[:3]
        v = {"func": int(d[3]), "params": d[4:]}
        self.angletypes.setdefault(i, {}).setdefault(j, {})[k] = v
        self.angletypes.setdefault(k, {}).setdefault(j, {})[i] = v

    def _dihedraltypes(self, d):
        i, j, k, l = d[:4]
        v = {"func": int(d[4]), "params": d[5:]}
        self.dihedraltypes.setdefault(i, {}).setdefault(j, {}).setdefault(k, {})[l] = v
        self.dihedraltypes.setdefault(l, {}).setdefault(k, {}).setdefault(j, {})[i] = v

    def _moleculetype(self, d):
        self.current_molecule = d[0]
        self.moleculetype[d[0]] = int(d[1])      # nrexcl

    def _atoms(self, d):
        at = TopAtom(atom_id=int(d[0]), atom_type=d[1], chain_idx=int(d[2]), chain_name=d[3], name=d[4], cgnr=int(d[5]),
                     charge=float(d[6]) if len(d) > 6 else None, mass=float(d[7]) if len(d) > 7 else None,
                     molecule_name=self.current_molecule)
        self.molecules_data[self.current_molecule].setdefault("atoms", collections.OrderedDict())[at.atom_id] = at

    def _tuple_section(self, name, arity):
        def parse(d):
            ids = tuple(int(x) for x in d[:arity])
            self.molecules_data[self.current_molecule].setdefault(name, collections.OrderedDict())[ids] = d[arity:]
        return parse

    def _system(self, d):
        self.system_name = " ".join(d)

    def _molecules(self, d):
        self.molecules.append((d[0], int(d[1])))

    def read(self):
        parsers = {"defaults": self._defaults, "atomtypes": self._atomtypes, "atomstate": self._atomstate,
                   "nonbond_params": self._nonbond_params, "bondtypes": self._bondtypes, "angletypes": self._angletypes,
                   "dihedraltypes": self._dihedraltypes, "moleculetype": self._moleculetype, "atoms": self._atoms,
                   "bonds": self._tuple_section("bonds", 2), "angles": self._tuple_section("angles", 3),
                   "dihedrals": self._tuple_section("dihedrals", 4), "improper_dihedrals": self._tuple_section("improper_dihedrals", 4),
                   "pairs": self._tuple_section("pairs", 2), "system": self._system, "molecules": self._molecules}
        current, section, previous = None, None, None
        for raw in preprocess_topology(self.file_name):
            line = re.sub(";.*$", "", raw.strip()).strip()
            if not line or line.startswith("#"):
                continue
            if line.startswith("["):
                previous, section = section, line.replace("[", "").replace("]", "").strip()
                if previous == "dihedrals" and section == "dihedrals":
                    section = "improper_dihedrals"
                current = parsers.get(section)
                continue
            if current is not None:
                d = line.split()
                if d:
                    current(d)
        return self


def read_exclusion_list(file_name):
    """`id1 id2` per line (start_simulation.py:174-187)."""
    with open(file_name) as f:
        return [tuple(int(x) for x in l.split()) for l in f if l.strip()]


def write_exclusion_list(file_name, exclusions):
    from ..rank import wopen
    with wopen(file_name, "w") as f:
        f.write("\n".join("%d %d" % tuple(d) for d in sorted(exclusions)))
