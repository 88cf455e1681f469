"""Command line of the driver: same flags, defaults and `@params` file syntax as the reference
(/root/reference/src/app_args.py:29-43,71-211).  `key=value` lines of a params file become
`--key=value`; python literals (True/False) are parsed with ast.literal_eval."""
import argparse
import ast
import random

_LIT = ast.literal_eval

# (flags, kwargs) in the reference's order; grouped as in app_args.py:71-211
ARGS = [
    # general
    (("--conf",), dict(required=True, help="input .gro coordinate file")),
    (("--top", "--topology"), dict(required=True, dest="top", help="GROMACS-like topology file")),
    (("--node_grid",), dict()),
    (("--skin",), dict(default=0.16)),                      # kept as given ("auto" allowed), like the reference
    (("--output_prefix",), dict(type=str, default="sim")),
    (("--output_file",), dict(type=str, default="trjout.h5")),
    (("--trj_collect",), dict(type=int, default=1000)),
    (("--energy_collect",), dict(type=int, default=1000)),
    (("--topol_collect",), dict(type=int, default=1000)),
    (("--reactions",), dict()),
    (("--debug",), dict()),
    (("--check_topology",), dict(type=_LIT, default=False)),
    (("--start_ar",), dict(type=int, default=0)),
    (("--stop_ar",), dict(type=int, default=-1)),
    (("--table_groups",), dict()),
    (("--max_force",), dict(type=float, default=-1)),
    (("--rate_arrhenius",), dict(type=_LIT, default=False)),
    (("--exclusion_list",), dict()),
    (("--benchmark_data",), dict()),
    (("--system_monitor_filter",), dict()),
    (("--do_not_exclude_bonds",), dict(type=_LIT, default=False)),
    (("--kb",), dict(type=float, default=0.0083144621)),
    (("--mass_factor",), dict(type=float, default=1.6605402)),
    # simulation
    (("--run",), dict(type=int, default=10000)),
    (("--int_step",), dict(type=int, default=1000)),
    (("--rng_seed",), dict(type=int, default=random.randint(1000, 10000))),   # random unless given, like the reference
    (("--thermal_groups",), dict()),
    (("--gen_velocity",), dict(type=_LIT, default=False)),
    (("--thermostat",), dict(default="lv", choices=("lv", "vr", "iso", "br", "no"))),   # 'no' added
    (("--barostat",), dict(default="lv", choices=("lv", "br"))),
    (("--barostat_tau",), dict(type=float, default=5.0)),
    (("--barostat_mass",), dict(type=float, default=50.0)),
    (("--barostat_gammaP",), dict(type=float, default=1.0)),
    (("--thermostat_gamma",), dict(type=float, default=5.0)),
    (("--temperature",), dict(type=float, default=458.0)),
    (("--pressure",), dict(type=float, default=None)),
    (("--dt",), dict(type=float, default=0.001)),
    (("--lj_cutoff",), dict(type=float, default=1.2)),
    (("--cg_cutoff",), dict(type=float, default=1.4)),
    (("--coulomb_epsilon1",), dict(type=float, default=1.0)),
    (("--coulomb_epsilon2",), dict(type=float, default=80.0)),
    (("--coulomb_kappa",), dict(type=float, default=0.0)),
    (("--coulomb_cutoff",), dict(type=float, default=0.9)),
    # H5MD storage
    (("--store_species",), dict(type=_LIT, default=True)),
    (("--store_state",), dict(type=_LIT, default=True)),
    (("--store_position",), dict(type=_LIT, default=True)),
    (("--store_lambda",), dict(type=_LIT, default=False)),
    (("--store_force",), dict(type=_LIT, default=False)),
    (("--store_velocity",), dict(type=_LIT, default=False)),
    (("--store_charge",), dict(type=_LIT, default=False)),
    (("--store_mass",), dict(type=_LIT, default=True)),
    (("--store_res_id",), dict(type=_LIT, default=True)),
    (("--store_pressure",), dict(type=_LIT, default=False)),
    (("--store_single_precision",), dict(type=_LIT, default=True)),
    (("--save_before_reaction",), dict(type=_LIT, default=False)),
    (("--trj_flush",), dict(type=int, default=None)),
    (("--gro_trj_collect",), dict(type=int, default=None)),
    (("--store_angdih",), dict(type=_LIT, default=False)),
    # maximum conversion
    (("--maximum_conversion",), dict()),
    (("--eq_steps",), dict(type=int, default=0)),
    (("--keep_simulation",), dict(default=False)),
    # counters
    (("--count_types",), dict()),
    (("--count_tuples",), dict(type=_LIT, default=False)),
    (("--count_types_state",), dict()),
    (("--count_fix_distances",), dict(type=_LIT, default=False)),
    # hybrid bonds
    (("--t_hybrid_bond",), dict(type=int, default=0)),
    (("--t_hybrid_angle",), dict(type=int, default=0)),
    (("--t_hybrid_dihedral",), dict(type=int, default=0)),
]


class MyArgParser(argparse.ArgumentParser):
    """`@file` expansion where every `key=value` line turns into `--key=value`; blank lines and
    `#` comments are skipped (app_args.py:33-42)."""

    def convert_arg_line_to_args(self, line):
        t = line.strip()
        if not t or t.startswith("#"):
            return []
        t = t.split("#")[0].strip()
        if "=" in t:
            k, v = t.split("=", 1)
            return ["--%s=%s" % (k.strip().lstrip("-"), v.strip())]
        return [t if t.startswith("--") else "--" + t]

    def save_to_file(self, output_file, namespace):
        from ..rank import wopen
        with wopen(output_file, "w") as f:
            for k, v in sorted(vars(namespace).items()):
                if v is not None:
                    f.write("%s=%s\n" % (k, v))


def _args():
    p = MyArgParser(description="Runs the reactive MD simulation", fromfile_prefix_chars="@")
    for flags, kw in ARGS:
        p.add_argument(*flags, **kw)
    return p
