"""Text outputs of a ChemLab run: end configurations (.gro), the tuple lists with their
parameters (`*_bonds.dat`, `*_angles.dat`, `*_dihedrals.dat`), the output topology (`*_output_topol.top`) and the
topology-manager dumps.

File layouts follow /root/reference/src/chemlab/files_io.py (GROFile.write :216-257, GROMACSTopologyFile.write
:534-607 with the section writers :822-960) and /root/reference/src/start_simulation.py:800-1036 (which rows go into
the .dat files and how the output topology is assembled).  H5MD (`io.DumpH5MD`, `DumpTopology`) needs h5py, which
this image does not have: trajectories are out of scope, the end state is not.

The three `TopologyManager.save_*` files are written by ESPResSo++ itself in the reference (C++ side, not under
/root/reference): the layout used here -- one `key: members...` row per entry, ascending -- is this build's own
[EXT-RECALL]."""
from ..rank import wopen
import collections
import os


def _prepare_path(file_name):
    d = os.path.dirname(file_name)
    if d and not os.path.isdir(d):
        os.makedirs(d, exist_ok=True)      # (several ranks may arrive here together)
    return file_name


def write_gro(file_name, conf, positions, box, velocities=None, title=None):
    """conf: files_io.GROFile (atom names / residue names come from the input file); positions, velocities: {atom_id: xyz}.
    Fixed columns `%5d%-5s%5s%5d%8.3f%8.3f%8.3f[%8.3f%8.3f%8.3f]`, box line `%f %f %f`, one trailing newline
    (files_io.py:232-254)."""
    out = [title or conf.title or "XXX of molecules", "%d" % len(conf.atoms)]
    for at_id in sorted(conf.atoms):
        at = conf.atoms[at_id]
        p = positions[at_id]
        if velocities is None:
            out.append("%5d%-5s%5s%5d%8.3f%8.3f%8.3f" % (at.chain_idx, at.chain_name, at.name, at.atom_id, p[0], p[1], p[2]))
        else:
            v = velocities[at_id]
            out.append("%5d%-5s%5s%5d%8.3f%8.3f%8.3f%8.3f%8.3f%8.3f" % (at.chain_idx, at.chain_name, at.name, at.atom_id,
                                                                         p[0], p[1], p[2], v[0], v[1], v[2]))
    out.append("%f %f %f\n" % tuple(box))
    with wopen(_prepare_path(file_name), "w") as f:
        f.write("\n".join(out))


TopoAtom = collections.namedtuple("TopoAtom", "atom_id atom_type chain_idx chain_name name cgnr charge mass atom_type_id")


def output_atoms(system, gt, valid_type_ids=None):
    """Per-particle rows of the output topology from the CURRENT particle properties (start_simulation.py:852-901):
    type/mass/charge/res_id as they are after the reactions, names from the input topology."""
    eng = system.engine
    ids = eng.get_state("ID")
    types, mass, res = eng.get_state("TYPE"), eng.get_state("MASS"), eng.get_state("RESID")
    q = system.storage.charges() if hasattr(system.storage, "charges") else None
    atoms = collections.OrderedDict()
    for k, pid in enumerate(ids.tolist()):
        t = int(types[k])
        if valid_type_ids and t not in valid_type_ids:
            continue
        charge = float(q[k]) if q is not None else gt.atoms.get(pid, {}).get("charge", 0.0)
        if pid in gt.atoms:
            d = gt.atoms[pid]
            atoms[pid] = TopoAtom(pid, gt.atomtype_atomsym[t], int(res[k]), d["chain_name"], d["name"], pid, charge, float(mass[k]), t)
        else:   # particles the input topology does not know (start_simulation.py:881-901)
            atoms[pid] = TopoAtom(pid, gt.atomtype_atomsym[t], int(res[k]), "CH%d" % int(res[k]), "X%d" % t, pid, charge, float(mass[k]), t)
    return atoms


def _typed_params(table, names):
    node = table
    for n in names:
        node = node.get(n) if isinstance(node, dict) else None
        if node is None:
            return None
    return node


def tuple_rows(kind, static_lists, dynamic_lists, chem_fpls, atoms, gt):
    """Rows of `*_bonds.dat` / `*_angles.dat` / `*_dihedrals.dat` (start_simulation.py:903-991).
    static_lists: [(fixed list, (func, [params]))]; dynamic_lists: [fixed list] whose parameters follow the CURRENT
    particle types through the topology's [ bondtypes ] / [ angletypes ] / [ dihedraltypes ]; chem_fpls: the reaction
    bond lists (bonds only), parameters from [ bondtypes ] of the current types as well."""
    table = {"bonds": gt.gt.bondtypes, "angles": gt.gt.angletypes, "dihedrals": gt.gt.dihedraltypes}[kind]
    get_all = {"bonds": "getAllBonds", "angles": "getAllTriples", "dihedrals": "getAllQuadruples"}[kind]
    rows = []
    for fl, (func, params) in static_lists:
        for p in getattr(fl, get_all)():
            rows.append(list(p) + [func] + list(params) + ["; static"])
    for fl in dynamic_lists:
        for p in getattr(fl, get_all)():
            names = [atoms[i].atom_type for i in p]
            prm = _typed_params(table, names) or _typed_params(table, names[::-1])
            if prm:
                rows.append(list(p) + [prm["func"]] + list(prm["params"]) + ["; dynamic"])
            else:
                rows.append(list(p) + ["; MISSING params type: %s dynamic" % "-".join(names)])
    if kind == "bonds":
        for fpl in chem_fpls:
            for p in fpl.getAllBonds():
                n0, n1 = atoms[p[0]].atom_type, atoms[p[1]].atom_type
                prm = _typed_params(table, (n0, n1)) or _typed_params(table, (n1, n0))
                if prm:
                    rows.append([p[0], p[1], "%s %s ; chem %s-%s" % (prm["func"], " ".join(prm["params"]), n0, n1)])
                else:
                    rows.append([p[0], p[1], "; chem MISSING params type: %s-%s" % (n0, n1)])
    return rows


def write_rows(file_name, rows):
    with wopen(_prepare_path(file_name), "w") as f:
        for r in rows:
            f.write("%s\n" % " ".join(str(x) for x in r))


def write_output_topology(file_name, gt, atoms, bonds, angles, dihedrals):
    """`*_output_topol.top`: one molecule type `MOL` (nrexcl 3) holding every particle, the force-field sections of
    the input topology, and the tuple lists as written to the .dat files (start_simulation.py:838-851,992-993;
    section order and row formats files_io.py:549-607,822-960)."""
    t = gt.gt
    sec = []

    def section(name, lines):
        sec.append("\n[ %s ]\n" % name)
        sec.extend("%s\n" % l for l in lines)
        sec.append("\n")

    if t.defaults:
        d = dict(t.defaults)
        section("defaults", ["%s %s %s %s %s" % (d["nbfunc"], d["combinationrule"], "yes" if d["gen-pairs"] else "no", d["fudgeLJ"], d["fudgeQQ"])])
    if t.atomtypes:
        section("atomtypes", ["{name} {mass} {charge} {type} {sigma} {epsilon}".format(**v) for v in t.atomtypes.values()])
    if t.bondtypes:
        section("bondtypes", ["%s %s %s %s" % (i, j, p["func"], " ".join(p["params"])) for i in t.bondtypes for j, p in t.bondtypes[i].items()])
    if t.angletypes:
        section("angletypes", ["%s %s %s %s %s" % (i, j, k, p["func"], " ".join(p["params"]))
                               for i in t.angletypes for j in t.angletypes[i] for k, p in t.angletypes[i][j].items()])
    if t.dihedraltypes:
        section("dihedraltypes", ["%s %s %s %s %s %s" % (i, j, k, l, p["func"], " ".join(p["params"]))
                                  for i in t.dihedraltypes for j in t.dihedraltypes[i] for k in t.dihedraltypes[i][j]
                                  for l, p in t.dihedraltypes[i][j][k].items()])
    if t.nonbond_params:
        section("nonbond_params", ["%s %s %s %s" % (k[0], k[1], p["func"], " ".join(str(x) for x in p["params"])) for k, p in t.nonbond_params.items()])
    if t.atomstate:
        section("atomstate", ["%s %s" % kv for kv in t.atomstate.items()])
    section("moleculetype", ["MOL 3"])
    section("atoms", ["%s %s %s %s %s %s %s %s" % (a.atom_id, a.atom_type, a.chain_idx, a.chain_name, a.name, a.cgnr,
                                                    a.charge if a.charge is not None else "0.0", a.mass if a.mass is not None else "")
                      for a in (atoms[k] for k in sorted(atoms))])
    for name, rows, arity in (("bonds", bonds, 2), ("angles", angles, 3), ("dihedrals", dihedrals, 4)):
        flat = sorted([list(r[:arity]) + [str(x) for x in r[arity:]] for r in rows], key=lambda r: r[:arity])
        section(name, [" ".join(str(x) for x in r) for r in flat])
    section("pairs", [])
    section("system", [t.system_name or "system"])
    section("molecules", ["MOL 1"])
    with wopen(_prepare_path(file_name), "w") as f:
        f.writelines(sec)


def bond_graph(fixed_pair_lists):
    adj = collections.defaultdict(set)
    for fpl in fixed_pair_lists:
        for a, b in fpl.getAllBonds():
            adj[a].add(b)
            adj[b].add(a)
    return adj


def write_topology_dumps(prefix, system, fixed_pair_lists):
    """TopologyManager.save_topology / save_res_topology / save_residues (start_simulation.py:1004-1006)."""
    eng = system.engine
    ids, res = eng.get_state("ID").tolist(), eng.get_state("RESID").tolist()
    res_of = dict(zip(ids, res))
    adj = bond_graph(fixed_pair_lists)
    with wopen(_prepare_path(prefix + "_topology.dat"), "w") as f:            # particle: bonded partners
        for pid in sorted(adj):
            f.write("%d: %s\n" % (pid, " ".join(str(x) for x in sorted(adj[pid]))))
    radj = collections.defaultdict(set)
    for a in adj:
        for b in adj[a]:
            if res_of[a] != res_of[b]:
                radj[res_of[a]].add(res_of[b])
    with wopen(_prepare_path(prefix + "_res_topology.dat"), "w") as f:        # residue: residues it is bonded to
        for r in sorted(radj):
            f.write("%d: %s\n" % (r, " ".join(str(x) for x in sorted(radj[r]))))
    members = collections.defaultdict(list)
    for pid, r in zip(ids, res):
        members[r].append(pid)
    with wopen(_prepare_path(prefix + "_residue_list.dat"), "w") as f:        # residue: its particles
        for r in sorted(members):
            f.write("%d: %s\n" % (r, " ".join(str(x) for x in sorted(members[r]))))
