"""Topology -> particle properties, bonded lists, exclusions and interactions.

Behaviour of /root/reference/src/chemlab/gromacs_topology.py for the in-scope func codes (SURVEY.md
Appendix B): LJ `func 1`, tabulated `func 8`, bonds harmonic `1` / FENE `7` / tabulated `8` / FENE+LJ `9`,
angles harmonic `1` / tabulated `8` / cosine `11`, dihedrals n-cosine `1` / Ryckaert-Bellemans `3` /
tabulated `8` / harmonic `12`, 1-4 pairs (LJ with fudgeLJ).  Type ids: first-seen order over [ molecules ], then the
remaining [ atomtypes ] in FILE order (the reference's py2 dict order is not reproducible
and type ids are internal labels only).  Exclusions: bonds + neighbours up to `nrexcl` bonds away,
with the atom-id offset per molecule type done correctly (the reference's offset bug is
not copied; single-molecule-type inputs give identical lists)."""
import collections
import math
import os

from . import files_io, tables


def convertc6c12(c6, c12, cr):
    """(c6, c12) -> (sigma, epsilon) for combination rule 1 (gromacs_topology.py:110-121)."""
    if cr == 1:
        if c6 == 0.0 or c12 == 0.0:
            return 0.0, 0.0
        sig = (c12 / c6) ** (1.0 / 6.0)
        return sig, c6 / (4.0 * sig ** 6)
    return c6, c12


def combination(sig_1, eps_1, sig_2, eps_2, cr):
    """rule 2: arithmetic sigma; otherwise geometric sigma; epsilon always geometric (:452-460)."""
    sig = 0.5 * (sig_1 + sig_2) if cr == 2 else math.sqrt(sig_1 * sig_2)
    return sig, math.sqrt(eps_1 * eps_2)


class GromacsTopology(object):
    def __init__(self, input_topol, generate_exclusions=True):
        self.gt = files_io.GROMACSTopologyFile(input_topol) if isinstance(input_topol, str) else input_topol
        self.generate_exclusions = generate_exclusions
        self.atoms = {}
        self.bonds, self.angles, self.dihedrals, self.pairs = [collections.OrderedDict() for _ in range(4)]
        self.exclusions = set()
        self.atomsym_atomtype = collections.OrderedDict()

    def read(self):
        if self.gt.defaults is None:
            self.gt.read()
        self._prepare_data()
        return self

    # molecule replication: ids are 1-based and contiguous
    def _prepare_data(self):
        gt = self.gt
        cr = gt.defaults["combinationrule"]
        offset = 0
        for mol_name, n_mols in gt.molecules:
            md = gt.molecules_data[mol_name]
            atoms = md.get("atoms", {})
            n_atoms = len(atoms)
            per_atom = {}
            for at_id in sorted(atoms):
                a = atoms[at_id]
                at = gt.atomtypes[a.atom_type]
                if a.atom_type not in self.atomsym_atomtype:
                    self.atomsym_atomtype[a.atom_type] = len(self.atomsym_atomtype)
                sig, eps = convertc6c12(at["sigma"], at["epsilon"], cr)
                per_atom[at_id] = {"type": a.atom_type, "type_id": self.atomsym_atomtype[a.atom_type], "sig": sig, "eps": eps,
                                   "state": at.get("state", 0), "charge": a.charge if a.charge else at["charge"],
                                   "mass": a.mass if a.mass else at["mass"], "name": a.name, "chain_name": a.chain_name,
                                   "chain_idx": a.chain_idx, "molecule_name": mol_name}
            for mol in range(n_mols):
                for k, v in per_atom.items():
                    self.atoms[offset + k + mol * n_atoms] = v
            for name, dst in (("bonds", self.bonds), ("angles", self.angles), ("dihedrals", self.dihedrals), ("pairs", self.pairs)):
                for ids, params in md.get(name, {}).items():
                    for mol in range(n_mols):
                        dst[tuple(offset + i + mol * n_atoms for i in ids)] = params
            if self.generate_exclusions and md.get("bonds"):
                mol_excl = generate_exclusions(list(md["bonds"].keys()), gt.moleculetype[mol_name])
                for mol in range(n_mols):
                    for l in mol_excl:
                        self.exclusions.add(tuple(sorted(offset + i + mol * n_atoms for i in l)))
            offset += n_mols * n_atoms
        for k, v in gt.nonbond_params.items():
            if v["func"] == 1 and cr == 1 and v["params"]:
                v["params"][0], v["params"][1] = convertc6c12(float(v["params"][0]), float(v["params"][1]), cr)
        for name in gt.atomtypes:        # remaining types (needed by reactions that create them)
            if name not in self.atomsym_atomtype:
                self.atomsym_atomtype[name] = len(self.atomsym_atomtype)
        self.used_atomsym_atomtype = self.atomsym_atomtype
        self.atomtype_atomsym = {v: k for k, v in self.atomsym_atomtype.items()}


def generate_exclusions(bond_list, nrexcl):
    """Pairs of one molecule separated by at most `nrexcl` bonds (plus the bonds themselves)."""
    adj = collections.defaultdict(set)
    for a, b in bond_list:
        adj[a].add(b)
        adj[b].add(a)
    excl = {tuple(sorted(b)) for b in bond_list}
    for root in adj:
        frontier, seen = {root}, {root}
        for _ in range(nrexcl):
            frontier = {n for f in frontier for n in adj[f]} - seen
            seen |= frontier
            for n in frontier:
                excl.add(tuple(sorted((root, n))))
    return excl


def gen_particle_list(coordinate, topol, espressopp):
    """props + particle rows (id, type, pos, mass, q, res_id, state, lambda_adr) (:1418-1441)."""
    props = ["id", "type", "pos", "mass", "q", "res_id", "state", "lambda_adr"]
    plist = []
    for atom_id in sorted(coordinate.atoms):
        d, t = coordinate.atoms[atom_id], topol.atoms[atom_id]
        plist.append([atom_id, t["type_id"], espressopp.Real3D(d.position), t["mass"], t["charge"], d.chain_idx, t.get("state", 0), 1.0])
    return props, plist


def set_nonbonded_interactions(espressopp, system, gt, vl, lj_cutoff, tab_cutoff=None, tables_=None, table_dir=".", cr_observs=None):
    """One VerletListLennardJones ('lj') and one VerletListTabulated ('lj-tab') for all type pairs
    (:463-899).  A pair gets no potential when sigma <= 0 (:715)."""
    tab_cutoff = lj_cutoff if tab_cutoff is None else tab_cutoff
    tables_ = tables_ or []
    cr = int(gt.gt.defaults["combinationrule"])
    lj = espressopp.interaction.VerletListLennardJones(vl)
    tab = espressopp.interaction.VerletListTabulated(vl)
    mix = espressopp.interaction.VerletListMixedTabulated(vl)
    cr_observs = {} if cr_observs is None else cr_observs
    has_lj = has_tab = has_mix = False

    def pot_file(name):                                  # table.xvg -> table.pot next to it, converted when missing (:764-769)
        xvg = os.path.join(table_dir, name)
        pot = xvg.replace(".xvg", "") + ".pot"
        if not os.path.exists(pot):
            tables.convert_table(xvg, pot)
        return pot
    names = list(gt.used_atomsym_atomtype)
    for i, n1 in enumerate(names):
        for n2 in names[i:]:
            t1, t2 = gt.used_atomsym_atomtype[n1], gt.used_atomsym_atomtype[n2]
            param = gt.gt.nonbond_params.get(tuple(sorted((n1, n2))))
            table_name, sig, eps = None, -1.0, -1.0
            if param:
                if param["func"] == 1:
                    sig, eps = float(param["params"][0]), float(param["params"][1])
                elif param["func"] == 8:
                    table_name = param["params"][0] if param["params"] else "table_%s_%s.xvg" % (n1, n2)
                elif param["func"] in (10, 12):
                    # func 10: `tab1 tab2 TYPE total` -- U = x tab1 + (1 - x) tab2 with x = (particles of TYPE) / total, the
                    # chemical conversion (:574-583,770-778); func 12: `tab1 tab2 x` with a constant x (:612-618,779-786)
                    pr = param["params"]
                    if param["func"] == 10:
                        cr_type, cr_total = gt.used_atomsym_atomtype[pr[2]], int(pr[3])
                        key = (cr_type, cr_total, None)
                        if key not in cr_observs:
                            cr_observs[key] = espressopp.analysis.ChemicalConversion(system, cr_type, cr_total)
                        pot = espressopp.interaction.MixedTabulated(1, pot_file(pr[0]), pot_file(pr[1]), cr_observs[key], cutoff=tab_cutoff)
                    else:
                        pot = espressopp.interaction.MixedTabulated(1, table1=pot_file(pr[0]), table2=pot_file(pr[1]), mix_value=float(pr[2]), cutoff=tab_cutoff)
                    mix.setPotential(type1=t1, type2=t2, potential=pot)
                    has_mix = True
                    continue
                else:
                    raise NotImplementedError("nonbond_params func %d is outside the hot-path scope (SURVEY.md 2.1 #3)" % param["func"])
            elif n1 in tables_ and n2 in tables_:
                table_name = "table_%s_%s.xvg" % (n1, n2)
            else:
                a1, a2 = gt.gt.atomtypes[n1], gt.gt.atomtypes[n2]
                s1, e1 = convertc6c12(a1["sigma"], a1["epsilon"], cr)
                s2, e2 = convertc6c12(a2["sigma"], a2["epsilon"], cr)
                sig, eps = combination(s1, e1, s2, e2, cr)
            if table_name is not None:
                xvg = os.path.join(table_dir, table_name)
                pot = xvg.replace(".xvg", "") + ".pot"
                if not os.path.exists(pot):
                    tables.convert_table(xvg, pot)
                tab.setPotential(type1=t1, type2=t2, potential=espressopp.interaction.Tabulated(itype=1, filename=pot, cutoff=tab_cutoff))
                has_tab = True
            elif sig > 0.0:
                lj.setPotential(type1=t1, type2=t2, potential=espressopp.interaction.LennardJones(epsilon=eps, sigma=sig, cutoff=lj_cutoff))
                has_lj = True
    if has_lj:
        system.addInteraction(lj, "lj")
    if has_tab:
        system.addInteraction(tab, "lj-tab")
    if has_mix:
        system.addInteraction(mix, "lj-mix_tab")
    return lj if has_lj else None, tab if has_tab else None


def _type_params(table, types_, default=None):
    node = table
    for t in types_:
        node = node.get(t) if isinstance(node, dict) else None
        if node is None:
            return default
    return node


def set_bonded_interactions(espressopp, system, gt, dynamic_type_ids=(), table_dir="."):
    """[ bonds ] -> FixedPairList interactions (:902-1061).  Entries whose type pair can change through
    a reaction go to ONE Types list (parameters by current types); the rest are grouped by parameters."""
    groups, dyn = collections.OrderedDict(), []
    typed_params = {}
    for (b1, b2), prm in gt.bonds.items():
        n1, n2 = gt.atoms[b1]["type"], gt.atoms[b2]["type"]
        spec = _type_params(gt.gt.bondtypes, (n1, n2)) if not prm else {"func": int(prm[0]), "params": prm[1:]}
        if spec is not None and not spec["params"]:
            spec = _type_params(gt.gt.bondtypes, (n1, n2))
        if spec is None:
            raise RuntimeError("no bond parameters for %s-%s" % (n1, n2))
        t1, t2 = gt.atoms[b1]["type_id"], gt.atoms[b2]["type_id"]
        if t1 in dynamic_type_ids or t2 in dynamic_type_ids:
            dyn.append((b1, b2))
        else:
            groups.setdefault((spec["func"], tuple(float(x) for x in spec["params"])), []).append((b1, b2))
    out = {}

    def pot_of(func, p):
        if func == 1:
            return espressopp.interaction.Harmonic(K=p[1] / 2.0, r0=p[0])       # K = k_gmx / 2 (:918)
        if func == 7:
            return espressopp.interaction.FENE(K=p[1], r0=0.0, rMax=p[0])
        if func == 8:                                                           # table_b<N>.xvg -> .pot if missing (:919-925)
            pot = os.path.join(table_dir, "table_b%d.pot" % int(p[0]))
            if not os.path.exists(pot):
                tables.convert_table(os.path.join(table_dir, "table_b%d.xvg" % int(p[0])), pot)
            return espressopp.interaction.Tabulated(itype=1, filename=pot)
        if func == 9:                                                           # rMax K sigma epsilon (:935-942)
            return espressopp.interaction.FENELennardJones(K=p[1], r0=0.0, rMax=p[0], sigma=p[2], epsilon=p[3])
        raise NotImplementedError("bond func %d is outside the hot-path scope" % func)
    for k, ((func, p), bl) in enumerate(groups.items()):
        fpl = espressopp.FixedPairList(system.storage)
        fpl.addBonds(bl)
        cls = {1: espressopp.interaction.FixedPairListHarmonic, 7: espressopp.interaction.FixedPairListFENE,
               8: espressopp.interaction.FixedPairListTabulated, 9: espressopp.interaction.FixedPairListFENELennardJones}.get(func)
        if cls is None:
            raise NotImplementedError("bond func %d is outside the hot-path scope" % func)
        fpl.params = (func, list(p))                       # what *_bonds.dat prints for a static list (:1049)
        inter = cls(system, fpl, pot_of(func, p))
        system.addInteraction(inter, "bond_%d" % k)
        out["bond_%d" % k] = (fpl, inter)
    if dyn:
        fpl = espressopp.FixedPairList(system.storage)
        fpl.addBonds(dyn)
        inter = espressopp.interaction.FixedPairListTypesHarmonic(system, fpl)
        for n1, row in gt.gt.bondtypes.items():
            for n2, spec in row.items():
                if spec["func"] == 1 and n1 in gt.used_atomsym_atomtype and n2 in gt.used_atomsym_atomtype:
                    p = [float(x) for x in spec["params"]]
                    inter.setPotential(gt.used_atomsym_atomtype[n1], gt.used_atomsym_atomtype[n2], pot_of(1, p))
        system.addInteraction(inter, "bond_dynamic")
        out["bond_dynamic"] = (fpl, inter)
    return out


def set_angle_interactions(espressopp, system, gt, dynamic_type_ids=(), table_dir="."):
    """[ angles ] -> FixedTripleList interactions (:1069-1176): func 1 AngularHarmonic(K=k/2, theta0 rad),
    func 11 Cosine(K, theta0 rad)."""
    groups, dyn = collections.OrderedDict(), []
    for ids, prm in gt.angles.items():
        names = [gt.atoms[i]["type"] for i in ids]
        spec = {"func": int(prm[0]), "params": prm[1:]} if prm and len(prm) > 1 else _type_params(gt.gt.angletypes, names)
        if spec is None:
            raise RuntimeError("no angle parameters for %s" % "-".join(names))
        if any(gt.atoms[i]["type_id"] in dynamic_type_ids for i in ids):
            dyn.append(ids)
        else:
            groups.setdefault((spec["func"], tuple(float(x) for x in spec["params"])), []).append(ids)

    def pot_of(func, p):
        if func == 1:
            return espressopp.interaction.AngularHarmonic(K=p[1] / 2.0, theta0=p[0] * math.pi / 180.0)
        if func == 11:
            return espressopp.interaction.Cosine(K=p[1], theta0=p[0] * math.pi / 180.0)
        if func == 8:                                                           # table_a<N>.xvg (degrees) -> .pot (radians) (:1074-1080)
            pot = os.path.join(table_dir, "table_a%d.pot" % int(p[0]))
            if not os.path.exists(pot):
                tables.convert_table(os.path.join(table_dir, "table_a%d.xvg" % int(p[0])), pot)
            return espressopp.interaction.TabulatedAngular(itype=1, filename=pot)
        raise NotImplementedError("angle func %d is outside the hot-path scope" % func)
    out = {}
    for k, ((func, p), tl) in enumerate(groups.items()):
        ftl = espressopp.FixedTripleList(system.storage)
        ftl.addTriples(tl)
        cls = {1: espressopp.interaction.FixedTripleListAngularHarmonic, 11: espressopp.interaction.FixedTripleListCosine,
               8: espressopp.interaction.FixedTripleListTabulatedAngular}.get(func)
        if cls is None:
            raise NotImplementedError("angle func %d is outside the hot-path scope" % func)
        ftl.params = (func, list(p))
        inter = cls(system, ftl, pot_of(func, p))
        system.addInteraction(inter, "angle_%d" % k)
        out["angle_%d" % k] = (ftl, inter)
    # one dynamic Types list also receives the angles that reactions spawn (TopologyManager.register_triplet)
    any_dyn_types = [(n1, n2, n3, spec) for n1, a in gt.gt.angletypes.items() for n2, b in a.items() for n3, spec in b.items()
                     if spec["func"] == 1 and all(n in gt.used_atomsym_atomtype for n in (n1, n2, n3))]
    if dyn or (dynamic_type_ids and any_dyn_types):
        ftl = espressopp.FixedTripleList(system.storage)
        ftl.addTriples(dyn)
        inter = espressopp.interaction.FixedTripleListTypesAngularHarmonic(system, ftl)
        for n1, n2, n3, spec in any_dyn_types:
            p = [float(x) for x in spec["params"]]
            ids = [gt.used_atomsym_atomtype[n] for n in (n1, n2, n3)]
            inter.setPotential(ids[0], ids[1], ids[2], pot_of(1, p))
        system.addInteraction(inter, "angle_dynamic")
        out["angle_dynamic"] = (ftl, inter)
    return out


def set_dihedral_interactions(espressopp, system, gt, dynamic_type_ids=(), table_dir="."):
    """[ dihedrals ] -> FixedQuadrupleList interactions (:1182-1308).  func 1 DihedralHarmonicNCos(K, phi0 rad,
    multiplicity) from `phi0 K n`; func 3 DihedralRB(K0..K5) from `C0..C5`; func 8 TabulatedDihedral(table_d<N>.pot);
    func 12 DihedralHarmonic(K, phi0 rad) from `phi0 K`.  Entries whose type tuple can change through a reaction go to
    ONE Types list per func, the rest are grouped by parameters.
    (The reference's func 3 conversion drops its first two coefficients, `t = raw[1:]; enumerate(t[1:])` at :1190-1191
    -- an indexing slip that is not copied: K_n = C_n here.)"""
    groups, dyn = collections.OrderedDict(), collections.OrderedDict()
    for ids, prm in gt.dihedrals.items():
        names = [gt.atoms[i]["type"] for i in ids]
        spec = {"func": int(prm[0]), "params": prm[1:]} if prm and len(prm) > 1 else _type_params(gt.gt.dihedraltypes, names)
        if spec is None:
            raise RuntimeError("no dihedral parameters for %s" % "-".join(names))
        if any(gt.atoms[i]["type_id"] in dynamic_type_ids for i in ids):
            dyn.setdefault(spec["func"], []).append(ids)
        else:
            groups.setdefault((spec["func"], tuple(float(x) for x in spec["params"])), []).append(ids)

    def pot_of(func, p):
        if func == 1:
            return espressopp.interaction.DihedralHarmonicNCos(K=p[1], phi0=p[0] * math.pi / 180.0, multiplicity=int(p[2]))
        if func == 3:
            return espressopp.interaction.DihedralRB(*[float(x) for x in p[:6]])
        if func == 8:
            pot = os.path.join(table_dir, "table_d%d.pot" % int(p[0]))
            if not os.path.exists(pot):
                tables.convert_table(os.path.join(table_dir, "table_d%d.xvg" % int(p[0])), pot)
            return espressopp.interaction.TabulatedDihedral(itype=1, filename=pot)
        if func == 12:
            return espressopp.interaction.DihedralHarmonic(K=p[1], phi0=p[0] * math.pi / 180.0)
        raise RuntimeError("Unknown func type")                                   # :1203
    static_cls = {1: "FixedQuadrupleListDihedralHarmonicNCos", 3: "FixedQuadrupleListDihedralRB",
                  8: "FixedQuadrupleListTabulatedDihedral", 12: "FixedQuadrupleListDihedralHarmonic"}
    typed_cls = {1: "FixedQuadrupleListTypesDihedralHarmonicNCos", 3: "FixedQuadrupleListTypesDihedralRB",
                 8: "FixedQuadrupleListTypesTabulatedDihedral", 12: "FixedQuadrupleListTypesDihedralHarmonic"}
    out = {}
    for k, ((func, p), ql) in enumerate(groups.items()):
        if func not in static_cls:
            raise RuntimeError("Unknown func type")
        fql = espressopp.FixedQuadrupleList(system.storage)
        fql.addQuadruples(ql)
        fql.params = (func, list(p))
        inter = getattr(espressopp.interaction, static_cls[func])(system, fql, pot_of(func, p))
        system.addInteraction(inter, "dihedral_%d" % k)
        out["dihedral_%d" % k] = (fql, inter)
    # dynamic Types list: also receives the dihedrals that reactions spawn (TopologyManager.register_quadruplet)
    typed = [(n1, n2, n3, n4, spec) for n1, a in gt.gt.dihedraltypes.items() for n2, b in a.items() for n3, c in b.items()
             for n4, spec in c.items() if all(n in gt.used_atomsym_atomtype for n in (n1, n2, n3, n4))]
    funcs = sorted(set(dyn) | ({spec["func"] for *_, spec in typed} if dynamic_type_ids else set()))
    for func in funcs[:1]:        # one dynamic list (the examples use one dihedral func per topology)
        fql = espressopp.FixedQuadrupleList(system.storage)
        fql.addQuadruples(dyn.get(func, []))
        inter = getattr(espressopp.interaction, typed_cls[func])(system, fql)
        for n1, n2, n3, n4, spec in typed:
            if spec["func"] != func:
                continue
            ids = [gt.used_atomsym_atomtype[n] for n in (n1, n2, n3, n4)]
            inter.setPotential(ids[0], ids[1], ids[2], ids[3], pot_of(func, [float(x) if func != 8 else x for x in spec["params"]]))
        system.addInteraction(inter, "dihedral_dynamic")
        out["dihedral_dynamic"] = (fql, inter)
    if len(funcs) > 1:
        raise NotImplementedError("dynamic dihedrals of more than one functional form in one topology")
    return out


def set_pair_interactions(espressopp, system, gt, lj_cutoff, dynamic_type_ids=()):
    """[ pairs ] (1-4 interactions) -> FixedPairList[Types]LennardJones (:1314-1411): explicit `sigma epsilon` on the
    pair line, else [ pairtypes ]-less gen-pairs: combination rule of the two atom types with epsilon * fudgeLJ.
    (The reference combines atom type 0 with itself, :1341-1342, and calls combination() with three arguments on the
    static path, :1360 -- SURVEY Q5; here both atom types are used.)"""
    if not gt.pairs:
        return {}
    cr = int(gt.gt.defaults["combinationrule"])
    fudge = float(gt.gt.defaults.get("fudgeLJ", 1.0))
    at = gt.gt.atomtypes
    groups, dyn = collections.OrderedDict(), []
    for (a, b), prm in gt.pairs.items():
        n1, n2 = gt.atoms[a]["type"], gt.atoms[b]["type"]
        if gt.atoms[a]["type_id"] in dynamic_type_ids or gt.atoms[b]["type_id"] in dynamic_type_ids:
            dyn.append((a, b))
            continue
        if prm and len(prm) > 2:
            sig, eps = float(prm[1]), float(prm[2])
        elif not gt.gt.defaults.get("gen-pairs"):
            # (the reference would look the pair up in gt.pairparams -- which nothing ever fills, :1337-1339 -- and then fail
            #  on a missing key, :1345: without gen-pairs a pair line must carry its parameters)
            raise RuntimeError("[ pairs ] entry %s-%s has no parameters and [ defaults ] gen-pairs is 'no'" % (a, b))
        else:
            s1, e1 = convertc6c12(at[n1]["sigma"], at[n1]["epsilon"], cr)
            s2, e2 = convertc6c12(at[n2]["sigma"], at[n2]["epsilon"], cr)
            sig, eps = combination(s1, e1, s2, e2, cr)
            eps *= fudge
        groups.setdefault((sig, eps), []).append((a, b))
    out = {}
    for k, ((sig, eps), bl) in enumerate(groups.items()):
        fpl = espressopp.FixedPairList(system.storage)
        fpl.addBonds(bl)
        fpl.params = (1, [sig, eps])
        inter = espressopp.interaction.FixedPairListLennardJones(system, fpl, espressopp.interaction.LennardJones(epsilon=eps, sigma=sig, cutoff=lj_cutoff))
        system.addInteraction(inter, "lj14_%d" % k)
        out["lj14_%d" % k] = (fpl, inter)
    if dyn:
        fpl = espressopp.FixedPairList(system.storage)
        fpl.addBonds(dyn)
        inter = espressopp.interaction.FixedPairListTypesLennardJones(system, fpl)
        names = list(gt.used_atomsym_atomtype)
        for i, n1 in enumerate(names):
            for n2 in names[i:]:
                t1, t2 = gt.used_atomsym_atomtype[n1], gt.used_atomsym_atomtype[n2]
                if t1 in dynamic_type_ids or t2 in dynamic_type_ids:
                    s1, e1 = convertc6c12(at[n1]["sigma"], at[n1]["epsilon"], cr)
                    s2, e2 = convertc6c12(at[n2]["sigma"], at[n2]["epsilon"], cr)
                    sig, eps = combination(s1, e1, s2, e2, cr)
                    if sig > 0:
                        inter.setPotential(t1, t2, espressopp.interaction.LennardJones(sigma=sig, epsilon=fudge * eps, cutoff=lj_cutoff))
        system.addInteraction(inter, "dyn_lj14")
        out["lj14_dynamic"] = (fpl, inter)
    return out
