#!/usr/bin/env python3
"""py3 driver with the control flow of ChemLab's start_simulation.py for the in-scope
configurations: same `@params` CLI, same cadence integers, same timer definition
(`integratorLoop` = wall seconds inside integrator.run) and the same 4-field benchmark.csv row.

    python -m chemlab_amd.start_simulation @params

Reference: /root/reference/src/start_simulation.py -- cadence :100-103,265-270,645-673; set-up order
:122-212; hooks.py discovery :214-228; maximum-conversion stop :280-287,759-777 (tools.py:102-180);
thermostat :329-354; main loop :728-796; text outputs :738-744,800-1036 (chemlab/outputs.py);
benchmark row :997-998.  H5MD trajectories (io.DumpH5MD / DumpTopology) need h5py, absent from this image.
"""
import math
import os
import re
import shutil
import sys
import time

from . import espp as espressopp
from . import rank
from .chemlab import app_args, files_io, gromacs_topology, outputs, reaction_parser, reaction_setup


def cadence(args, cr_interval=None):
    """The integers the outer loop runs on (start_simulation.py:100-103,265-270,645-673).
    Python-2 integer division of `run / integrator_step` is kept."""
    integrator_step = args.int_step
    if args.trj_collect > 0:
        integrator_step = min(args.int_step, args.trj_collect)
    topol_collect = args.topol_collect
    if cr_interval:
        integrator_step = min(cr_interval, integrator_step)
        topol_collect = min(cr_interval, args.topol_collect)
    sim_step = args.run // integrator_step
    has_reaction = bool(cr_interval)
    k_enable = int(math.ceil(args.start_ar / float(integrator_step))) if (args.start_ar >= 0 and has_reaction) else -1
    k_stop = int(math.ceil(args.stop_ar / float(integrator_step))) if (args.stop_ar >= 0 and has_reaction) else -1
    trj_collect = min(args.trj_collect, cr_interval) if cr_interval else args.trj_collect
    k_trj_collect = int(math.ceil(trj_collect / float(integrator_step)))
    if args.trj_flush is None:
        k_trj_flush = 25 if 25 < 10 * k_trj_collect else 10 * k_trj_collect
    else:
        k_trj_flush = int(math.ceil(args.trj_flush / float(integrator_step)))
    energy_collect = min(cr_interval, args.energy_collect) if cr_interval else args.energy_collect
    return dict(integrator_step=integrator_step, sim_step=sim_step, k_enable_reactions=k_enable, k_stop_reactions=k_stop,
                k_trj_collect=k_trj_collect, k_trj_flush=k_trj_flush, topol_collect=topol_collect, energy_collect=energy_collect)


def load_hooks(path="hooks.py"):
    """hooks.py in the working directory (start_simulation.py:214-228): a python file defining any of
    hook_init_reaction, hook_postsetup_reaction, hook_at_step, hook_before_sim, hook_end.  `import espressopp`
    inside it resolves to this package's shim."""
    if not os.path.exists(path):
        return {}
    sys.modules.setdefault("espressopp", espressopp)
    ns = {"__name__": "hooks", "espressopp": espressopp}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)
    return {k: ns[k] for k in ("hook_init_reaction", "hook_postsetup_reaction", "hook_at_step", "hook_before_sim", "hook_end") if k in ns}


def get_maximum_conversion(args, system, chem_fpls, gt):
    """--maximum_conversion "<what>:<max>:<total>,..." -> [(observable, stop value)] (tools.py:102-180):
    `T1-T2` -> number of reaction bonds between those types >= max; `T(s)+U(t)` / `T` / `T(s)` -> fraction of the
    `total` particles that have that type (and state) >= max/total."""
    out = []
    re_ts = re.compile(r"(?P<type>[A-Za-z0-9-]+)\(?(?P<state>\d?)\)?")
    for o in args.maximum_conversion.split(","):
        sym, max_number, tot_number = o.split(":")
        max_number, tot_number = int(max_number), int(tot_number)
        if "-" in sym:          # the bond list of the group whose reactions involve this type pair (tools.py:128-138)
            s1, s2 = re_ts.match(sym).groupdict()["type"].split("-")
            t1, t2 = gt.used_atomsym_atomtype[s1], gt.used_atomsym_atomtype[s2]
            for _, fpl, _ in chem_fpls:
                tl = getattr(fpl, "type_list", None)
                if tl is None or (t1, t2) in tl or (t2, t1) in tl:
                    out.append((espressopp.analysis.NFixedPairListEntries(system, fpl), max_number))
                    break
        elif "+" in sym:
            parts = []
            for ts in sym.split("+"):
                m = re_ts.match(ts).groupdict()
                parts.append(espressopp.analysis.ChemicalConversionTypeState(system, gt.used_atomsym_atomtype[m["type"]],
                                                                              int(m["state"]) if m["state"] else None, tot_number))
            class _Sum(object):
                def compute(self_inner):
                    return sum(p.compute() for p in parts)
            out.append((_Sum(), float(max_number) / tot_number))
        else:
            m = re_ts.match(sym).groupdict()
            tid = gt.used_atomsym_atomtype[m["type"]]
            obs = (espressopp.analysis.ChemicalConversionTypeState(system, tid, int(m["state"]), tot_number) if m["state"]
                   else espressopp.analysis.ChemicalConversion(system, tid, tot_number))
            out.append((obs, float(max_number) / tot_number))
    return out


def _conf_positions(system, conf, unfolded):
    eng = system.engine
    ids = eng.get_state("ID").tolist()
    pos = eng.get_state("POS_UNFOLDED" if unfolded else "POS")
    vel = eng.get_state("VEL")
    return ({pid: pos[k] for k, pid in enumerate(ids) if pid in conf.atoms},
            {pid: vel[k] for k, pid in enumerate(ids) if pid in conf.atoms})


def write_final_outputs(args, system, gt, conf, bonded, angles, dihedrals, chem_fpls, topology_manager, ar, reaction_index):
    """Everything start_simulation.py:836-1036 writes after the loop that does not need HDF5."""
    prefix = "%s_%s" % (args.output_prefix, args.rng_seed)
    valid = [gt.atomsym_atomtype[x] for x in args.table_groups.split(",")] if args.table_groups else None
    atoms = outputs.output_atoms(system, gt, valid)

    def split(groups):
        static, dynamic = [], []
        for name, (fl, inter) in groups.items():
            if name.endswith("_dynamic"):
                dynamic.append(fl)
            else:
                static.append((fl, inter.params))
        return static, dynamic
    b_static, b_dyn = split(bonded)
    a_static, a_dyn = split(angles)
    d_static, d_dyn = split(dihedrals)
    brows = outputs.tuple_rows("bonds", b_static, b_dyn, [f for _, f, _ in chem_fpls], atoms, gt)
    arows = outputs.tuple_rows("angles", a_static, a_dyn, [], atoms, gt)
    drows = outputs.tuple_rows("dihedrals", d_static, d_dyn, [], atoms, gt)
    outputs.write_rows(prefix + "_bonds.dat", brows)
    outputs.write_rows(prefix + "_angles.dat", arows)
    outputs.write_rows(prefix + "_dihedrals.dat", drows)
    outputs.write_output_topology(prefix + "_output_topol.top", gt, atoms, brows, arows, drows)
    topology_manager.save_topology(prefix + "_topology.dat")
    topology_manager.save_res_topology(prefix + "_res_topology.dat")
    topology_manager.save_residues(prefix + "_residue_list.dat")
    pos, _ = _conf_positions(system, conf, unfolded=False)
    outputs.write_gro(prefix + "_confout.gro", conf, pos, conf.box)
    espressopp.io.DumpGRO(system, system.integrator, filename=prefix + "_whole_confout.gro").dump()
    if ar is not None:
        ar.save_reaction_counters(prefix + "_reaction_counters")
        with rank.wopen(prefix + "_reaction_counters", "a") as f:
            f.write("\n\nReaction index\n")
            for ridx in sorted(reaction_index):
                f.write("%s %s\n" % (ridx, reaction_index[ridx]))
        ar.save_intra_inter_counter(prefix + "_intra_inter_counters")


def join_ranks(precision=32):
    """Started as `python -m torch.distributed.run --nproc-per-node N -m chemlab_amd.start_simulation @params` (the reference:
    `mpirun -n N`, node grid from the communicator size, start_simulation.py:152-163): every rank runs the driver, its engine
    is joined to the slab decomposition (RCCL over xGMI; hipIpc where ranks share a device), rank 0 writes the outputs.
    Returns (rank, world); (0, 1) when started plainly."""
    world, rk = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world <= 1:
        return 0, 1
    from . import multigpu
    eng, _ = multigpu.make_engine(rk, int(os.environ.get("LOCAL_RANK", str(rk))), world, precision)
    espressopp.set_engine_factory(lambda: eng)
    rank.set_root(rk == 0)
    return rk, world


def main(argv=None, hooks=None, quiet=False, ranks=None):
    rk, world = ranks if ranks is not None else (0, 1)
    quiet = quiet or rk != 0
    log = (lambda *a: None) if quiet else print
    parser = app_args._args()
    args = parser.parse_args(argv)
    parser.save_to_file("%sparams.out" % args.output_prefix, args)
    if args.pressure is not None:
        raise NotImplementedError("barostats are outside the MI355X hot-path scope")
    if args.coulomb_cutoff > 0:
        log("note: coulomb_cutoff=%g ignored -- truncated Coulomb is outside the hot-path scope" % args.coulomb_cutoff)
    kb, mass_factor = args.kb, args.mass_factor
    lj_cutoff, cg_cutoff = args.lj_cutoff, args.cg_cutoff
    max_cutoff = max(lj_cutoff, cg_cutoff)            # ignores coulomb_cutoff
    gt = gromacs_topology.GromacsTopology(args.top).read()
    conf = files_io.GROFile(args.conf).read()
    box = conf.box
    cad = cadence(args)
    skin = 0.16 if args.skin == "auto" else float(args.skin)
    rng_seed = args.rng_seed
    props, plist = gromacs_topology.gen_particle_list(conf, gt, espressopp)
    for p in plist:
        p[3] *= mass_factor
    log("Reads %d particles" % len(plist))
    if args.gen_velocity:
        vx, vy, vz = espressopp.tools.velocities.gaussian(args.temperature, len(plist), [p[3] for p in plist], kb=kb, seed=rng_seed)
        props.append("v")
        for i, p in enumerate(plist):
            p.append(espressopp.Real3D(vx[i], vy[i], vz[i]))
    system = espressopp.System()
    system.rng = espressopp.esutil.RNG(rng_seed)
    system.skin = skin
    node_grid = [int(x) for x in args.node_grid.split(",")] if args.node_grid else espressopp.tools.decomp.nodeGrid(world)
    cell_grid = espressopp.tools.decomp.cellGrid(box, node_grid, max_cutoff, skin)
    log("Cell grid: (%d, %d, %d)" % tuple(int(c) for c in cell_grid))
    system.bc = espressopp.bc.OrthorhombicBC(system.rng, box)
    system.storage = espressopp.storage.DomainDecomposition(system, node_grid, cell_grid)
    integrator = espressopp.integrator.VelocityVerlet(system)
    integrator.dt = args.dt
    system.integrator = integrator
    system.storage.addParticles(plist, *props)
    system.storage.decompose()
    if args.exclusion_list and os.path.exists(args.exclusion_list):
        gt.exclusions = set(files_io.read_exclusion_list(args.exclusion_list))
    if gt.exclusions:
        files_io.write_exclusion_list("exclusion_%s.list" % os.path.basename(args.top).split(".")[0], gt.exclusions)
    dynamic_exclusion_list = espressopp.DynamicExcludeList(integrator, sorted(gt.exclusions))
    log("Excluded pairs from LJ interaction: %d" % len(gt.exclusions))
    verletlist = espressopp.VerletList(system, cutoff=max_cutoff, exclusionlist=dynamic_exclusion_list)
    log("Bonds: %d\nAngles: %d\nDihedrals: %d" % (len(gt.bonds), len(gt.angles), len(gt.dihedrals)))
    topology_manager = espressopp.integrator.TopologyManager(system)
    system.topology_manager = topology_manager
    file_hooks = load_hooks()                     # hooks.py of the working directory; explicit `hooks` win
    if file_hooks:
        log("Found hooks.py")
    hooks = dict(file_hooks, **(hooks or {}))
    ar, chem_fpls, cr_interval, dynamic_types, integrator_extensions = None, [], None, set(), []
    reaction_index = {}
    if args.reactions is not None and os.path.exists(args.reactions):
        rc = reaction_parser.parse_config(args.reactions)
        sc = reaction_setup.SetupReactions(espressopp, system, verletlist, gt, topology_manager, rc, args)
        ar, chem_fpls = sc.setup_reactions()
        integrator_extensions = sc.extensions_to_integrator
        dynamic_types = sc.dynamic_types
        reaction_index = {k: cr["equation"] for k, cr in enumerate(c for g in rc["reactions"].values() for c in g["reaction_list"])}
        system.engine.set_option("count_intra_inter", 1)   # ar.save_intra_inter_counter at the end of the run
        if rank.is_root():
            shutil.copyfile(args.reactions, "%s_%s_%s" % (args.output_prefix, rng_seed, os.path.basename(args.reactions)))
        cr_interval = rc["general"]["interval"]
        cad = cadence(args, cr_interval)
        log("Change integrator step to %d" % cad["integrator_step"])
    table_groups = args.table_groups.split(",") if args.table_groups else []
    cr_observs = {}                                       # conversion observables of mixed tables (start_simulation.py:298-299)
    gromacs_topology.set_nonbonded_interactions(espressopp, system, gt, verletlist, lj_cutoff, tab_cutoff=cg_cutoff, tables_=table_groups,
                                                table_dir=os.path.dirname(os.path.abspath(args.top)), cr_observs=cr_observs)
    bonded = gromacs_topology.set_bonded_interactions(espressopp, system, gt, dynamic_types,
                                                      table_dir=os.path.dirname(os.path.abspath(args.top)))
    angles = gromacs_topology.set_angle_interactions(espressopp, system, gt, dynamic_types,
                                                     table_dir=os.path.dirname(os.path.abspath(args.top)))
    dihedrals = gromacs_topology.set_dihedral_interactions(espressopp, system, gt, dynamic_types,
                                                           table_dir=os.path.dirname(os.path.abspath(args.top)))
    pairs14 = gromacs_topology.set_pair_interactions(espressopp, system, gt, lj_cutoff, dynamic_types)   # start_simulation.py:308-309
    if args.max_force > -1:                               # start_simulation.py:320-324, before the thermostat
        integrator.addExtension(espressopp.integrator.CapForce(system, args.max_force))
        log("Cap force to %s" % args.max_force)
    # thermostat (start_simulation.py:329-354)
    temperature = args.temperature * kb
    if args.thermostat == "lv":
        th = espressopp.integrator.LangevinThermostat(system)
        th.temperature, th.gamma = temperature, args.thermostat_gamma
        integrator.addExtension(th)
    elif args.thermostat == "vr":                         # start_simulation.py:337-340
        th = espressopp.integrator.StochasticVelocityRescaling(system)
        th.temperature, th.coupling = temperature, args.thermostat_gamma
        integrator.addExtension(th)
    elif args.thermostat == "br":                         # start_simulation.py:341-344
        th = espressopp.integrator.BerendsenThermostat(system)
        th.temperature, th.tau = temperature, args.thermostat_gamma
        integrator.addExtension(th)
    elif args.thermostat == "iso":                        # start_simulation.py:345-348
        th = espressopp.integrator.Isokinetic(system)
        th.temperature, th.coupling = temperature, int(args.thermostat_gamma)
        integrator.addExtension(th)
    elif args.thermostat != "no":
        raise Exception("Wrong thermostat keyword: `%s`" % args.thermostat)   # start_simulation.py:351-352
    # topology manager wiring (start_simulation.py:378-441): spawned angles land in the dynamic Types list
    for name, (fl, inter) in list(bonded.items()):
        dynamic_exclusion_list.observe_tuple(fl)
        topology_manager.observe_tuple(fl)
    for _, fpl, _ in chem_fpls:
        dynamic_exclusion_list.observe_tuple(fpl)
        topology_manager.observe_tuple(fpl)
    if "angle_dynamic" in angles:
        ftl, inter = angles["angle_dynamic"]
        dynamic_exclusion_list.observe_triple(ftl)
        for types_ in inter._typed:
            topology_manager.register_triplet(ftl, *types_)
    if "dihedral_dynamic" in dihedrals:                   # start_simulation.py:428-441
        fql, inter = dihedrals["dihedral_dynamic"]
        dynamic_exclusion_list.observe_quadruple(fql)
        for types_ in inter._typed:
            topology_manager.register_quadruplet(fql, *types_)
    topology_manager.initialize_topology()
    integrator.addExtension(topology_manager)
    # observables (start_simulation.py:447-569)
    energy_file = "%s_energy_%s.csv" % (args.output_prefix, rng_seed)
    mon = espressopp.analysis.SystemMonitor(system, integrator, espressopp.analysis.SystemMonitorOutputCSV(energy_file))
    mon.add_observable("T", espressopp.analysis.Temperature(system))
    mon.add_observable("Ekin", espressopp.analysis.KineticEnergy(system))
    for k in range(system.getNumberOfInteractions()):
        mon.add_observable(system.getNameOfInteraction(k), espressopp.analysis.PotentialEnergy(system, system.getInteraction(k)), False)
    for i, (gname, fpl, _) in enumerate(chem_fpls):
        mon.add_observable("count_%d" % i, espressopp.analysis.NFixedPairListEntries(system, fpl))
    for (cr_type, cr_total, _), obs in sorted(cr_observs.items(), key=lambda kv: kv[0][:2]):
        # computed with every energy row: that is also what moves the mixed (func 10) tables along with the conversion
        mon.add_observable("cr_%s_%s" % (cr_type, cr_total), obs)
    integrator.addExtension(espressopp.integrator.ExtAnalyze(mon, cad["energy_collect"]))
    # trajectory + topology writer (start_simulation.py:571-657): H5MD tree, see espp._DumpH5MD for the on-disk form
    traj_file = espressopp.io.DumpH5MD(
        system, "%s_%s_traj.h5" % (args.output_prefix, rng_seed), group_name="atoms", static_box=True,
        store_species=args.store_species, store_res_id=args.store_res_id, store_charge=args.store_charge, store_position=args.store_position,
        store_state=args.store_state, store_lambda=args.store_lambda, store_force=args.store_force, store_velocity=args.store_velocity,
        store_mass=args.store_mass, is_single_prec=args.store_single_precision, chunk_size=256)
    dump_topol = espressopp.io.DumpTopology(system, integrator, traj_file)
    for i, (gname, fpl, _) in enumerate(chem_fpls):
        dump_topol.observe_tuple(fpl, "chem_bonds_%d" % i)                     # :592-593, read back by Checkup.ipynb / analyze.py
    for kind, lists, add_static, observe, label in (("bond", bonded, dump_topol.add_static_tuple, dump_topol.observe_tuple, "bonds"),
                                                    ("angle", angles, dump_topol.add_static_triple, dump_topol.observe_triple, "angles"),
                                                    ("dihedral", dihedrals, dump_topol.add_static_quadruple, dump_topol.observe_quadruple, "dihedrals")):
        for cnt, (name, (fl, _inter)) in enumerate(sorted(lists.items())):     # :595-642
            if name.endswith("_dynamic") and args.store_angdih:
                observe(fl, "dynamic_%s_%d" % (label, cnt))
            else:
                add_static(fl, "%s_%d" % (label, cnt))
    save_traj_topology = args.save_before_reaction if cad["k_enable_reactions"] > 1 else True     # :649
    ext_dump_added = False
    if args.topol_collect > 0 and save_traj_topology:
        integrator.addExtension(espressopp.integrator.ExtAnalyze(dump_topol, cad["topol_collect"]))
        ext_dump_added = True
        dump_topol.dump(); dump_topol.update()
    espressopp.analysis.CMVelocity(system).reset()
    maximum_conversion, eq_run = [], 0                    # start_simulation.py:280-287
    if args.maximum_conversion and ar is not None:
        maximum_conversion = get_maximum_conversion(args, system, chem_fpls, gt)
        if args.eq_steps > 0:
            eq_run = int(args.eq_steps / cad["sim_step"])
    # main loop (start_simulation.py:728-796)
    reactions_enabled = False
    stop_simulation = False
    total_time = time.time()
    integrator_loop = 0.0
    if "hook_before_sim" in hooks:
        hooks["hook_before_sim"](system, integrator, ar, gt)
    for k in range(cad["sim_step"]):
        mon.info() if not quiet else None
        if save_traj_topology and cad["k_trj_collect"] > 0 and k % cad["k_trj_collect"] == 0:      # start_simulation.py:730-731
            traj_file.dump(k * cad["integrator_step"], k * cad["integrator_step"] * args.dt)
        if save_traj_topology and cad["k_trj_flush"] > 0 and k % cad["k_trj_flush"] == 0:          # :732-734
            dump_topol.update()
            traj_file.flush()
        if cad["k_enable_reactions"] == k and ar is not None:
            log("Enabling chemical reactions")
            integrator.addExtension(ar)
            for ext in integrator_extensions:                # start_simulation.py:738-740 (ATRPActivator)
                integrator.addExtension(ext)
            reactions_enabled = True
            pos, vel = _conf_positions(system, conf, unfolded=True)      # start_simulation.py:741-745
            outputs.write_gro("%s_%s_before_reaction_confout.gro" % (args.output_prefix, args.rng_seed), conf, pos, conf.box, velocities=vel)
            if "hook_init_reaction" in hooks and not hooks["hook_init_reaction"](system, integrator, ar, gt, args):
                raise RuntimeError("hook_init_reaction return False")
            if not save_traj_topology:                                                             # start_simulation.py:750-757
                save_traj_topology = True
                if not ext_dump_added:
                    integrator.addExtension(espressopp.integrator.ExtAnalyze(dump_topol, cad["topol_collect"]))
                    ext_dump_added = True
                dump_topol.dump(); dump_topol.update()
        if reactions_enabled:
            if not stop_simulation:                        # start_simulation.py:759-770
                for obs, stop_value in maximum_conversion:
                    val = obs.compute()
                    if val >= stop_value:
                        log("Reaches %s of the conversion => Stop simulation" % val)
                        stop_simulation = True
            if stop_simulation:
                if eq_run == 0:
                    break
                eq_run -= 1
            if cad["k_stop_reactions"] == k or stop_simulation:
                ar.disconnect()
        t0 = time.time()
        integrator.run(cad["integrator_step"])
        integrator_loop += time.time() - t0
        if "hook_at_step" in hooks:
            hooks["hook_at_step"](system, integrator, ar, gt, args, k * cad["integrator_step"])
    total_time = time.time() - total_time
    if "hook_end" in hooks:
        hooks["hook_end"](system, integrator, ar, gt, args)
    mon.info() if not quiet else None
    nsteps_total = cad["sim_step"] * cad["integrator_step"]                                        # start_simulation.py:802-807
    traj_file.dump(nsteps_total, nsteps_total * args.dt)
    dump_topol.dump(); dump_topol.update()
    traj_path = traj_file.close()
    write_final_outputs(args, system, gt, conf, bonded, angles, dihedrals, chem_fpls, topology_manager, ar, reaction_index)
    for ext in integrator_extensions:
        if hasattr(ext, "save_stats"):
            ext.save_stats()                                  # ATRPActivator.stats_filename (reaction_post_process.py:390-396)
    npart = espressopp.analysis.NPart(system).compute()
    with rank.wopen("%s_%s_benchmark.csv" % (args.output_prefix, rng_seed), "a+") as f:
        f.write("%d %d %s %s\n" % (1, npart, total_time, integrator_loop))
    log("finished: %d steps, integratorLoop %.3f s, %.1f steps/s" % (cad["sim_step"] * cad["integrator_step"], integrator_loop,
                                                                       cad["sim_step"] * cad["integrator_step"] / max(integrator_loop, 1e-12)))
    return dict(system=system, integrator=integrator, gt=gt, ar=ar, chem_fpls=chem_fpls, cadence=cad, args=args, trajectory=traj_path,
                total_time=total_time, integrator_loop=integrator_loop, bonded=bonded, angles=angles, dihedrals=dihedrals, monitor=mon,
                stopped_by_conversion=stop_simulation)


if __name__ == "__main__":
    main(sys.argv[1:], ranks=join_ranks())
