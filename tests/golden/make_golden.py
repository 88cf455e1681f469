#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ from the reference checkout.  Run ONLY in the build
container (it reads /root/reference); the committed outputs are what the tests use.

What is produced and why:
  app_args_*.json            args namespace of the reference's own CLI parser (src/app_args.py, importable
                             under py3) on each example's `params` file           -> pins chemlab/app_args.py
  table_nb_excerpt.{xvg,pot} first rows of a GROMACS non-bonded table and the reference converter's output
  table_b1_excerpt.{xvg,pot} same for a bonded table (src/tests/table_b1.xvg)    -> pins chemlab/tables.py
  table_a5_excerpt.{xvg,pot} same for an angle table (examples/atrp_activator/table_a5.xvg): degrees -> radians
  setup_known_answers.json   counts / cell grid / type order printed in examples/atrp_lj/single
  data files (inputs, not source): src/tests/{topol.top,*.itp}, examples/atrp_lj/*, examples/chain_growth_catalytic/*,
                             examples/mf/espp_cg_1/{conf.gro,topol.top,params,reaction.cfg,table_A_A.xvg}
"""
import json
import os
import shutil
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, os.path.join(REF, "src"))
    sys.path.insert(0, os.path.join(REF, "tools"))
    import app_args                     # reference CLI (py2/3 clean)
    import convert_gromacs2espp as conv  # reference table converter

    for name, rel in [("atrp_lj", "examples/atrp_lj"), ("chain_growth_catalytic", "examples/chain_growth_catalytic"),
                      ("pccg_lj", "examples/pccg_lj/chemical_reactions"), ("mf_espp_cg_1", "examples/mf/espp_cg_1")]:
        ns = app_args._args().parse_args(["@" + os.path.join(REF, rel, "params")])
        with open(os.path.join(HERE, "app_args_%s.json" % name), "w") as f:
            json.dump(vars(ns), f, indent=1, sort_keys=True)

    def excerpt(src, dst, nrows):
        lines = open(src).read().splitlines(True)
        head = [l for l in lines if l.strip() and l.strip()[0] in "#@"]
        body = [l for l in lines if l.strip() and l.strip()[0] not in "#@"][:nrows]
        open(dst, "w").writelines(head + body)
    for stem, src in [("table_nb_excerpt", os.path.join(REF, "examples/mf/espp_cg_1_water/table_A_A.xvg")),
                      ("table_b1_excerpt", os.path.join(REF, "src/tests/table_b1.xvg")),
                      # angle table: the converter decides by the `_a<N>` in the file name (degrees -> radians, theta = 0 dropped)
                      ("table_a5_excerpt", os.path.join(REF, "examples/atrp_activator/table_a5.xvg"))]:
        xvg = os.path.join(HERE, stem + ".xvg")
        excerpt(src, xvg, 50)
        conv.convertTable(xvg, os.path.join(HERE, stem + ".pot"), 1, 1, 1, 1)

    copies = {"src_tests": ["src/tests/topol.top", "src/tests/diol_cg.itp", "src/tests/ter_cg.itp"],
              "atrp_lj": ["examples/atrp_lj/%s" % f for f in ("conf.gro", "topol.top", "ffnb.itp", "exclusion_topol.list", "params", "atrp.cfg")],
              "chain_growth_catalytic": ["examples/chain_growth_catalytic/%s" % f for f in ("conf.gro", "topol.top", "params", "reaction.cfg")],
              # 1000-bead melt with a GROMACS non-bonded table (nonbond_params func 8, table_groups=A) and harmonic reaction bonds
              "mf_espp_cg_1": ["examples/mf/espp_cg_1/%s" % f for f in ("conf.gro", "topol.top", "params", "reaction.cfg", "table_A_A.xvg")]}
    for sub, files in copies.items():
        os.makedirs(os.path.join(HERE, sub), exist_ok=True)
        for f in files:
            shutil.copyfile(os.path.join(REF, f), os.path.join(HERE, sub, os.path.basename(f)))

    # examples/atrp_lj/single:30-48,198-205 (a captured run of the atrp_activator inputs)
    ka = dict(source="examples/atrp_lj/single", particles=6000, box=13.40248, cutoff_max=2.0, skin=0.1, cell_grid=[6, 6, 6],
              excluded_pairs=6000, bonds=4000, angles=2000, dihedrals=0,
              # test_topology_reader.py:34-69: 1000 DIO x 1 atom + 1000 TER x 3 atoms
              src_tests_topol=dict(atoms=4000, bonds=2000, angles=1000),
              # test_reaction_parser.py:29-51 is an exchange equation (out of scope); the normal-reaction grammar is pinned by the shipped configs
              cadence=dict(chain_growth_catalytic=dict(int_step=500, trj_collect=1000, interval=500, integrator_step=500, sim_step=10, k_enable_reactions=4),
                           atrp_lj=dict(int_step=1000, trj_collect=1000, interval=200, integrator_step=200, sim_step=10, k_enable_reactions=2)))
    with open(os.path.join(HERE, "setup_known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
