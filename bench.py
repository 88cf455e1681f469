#!/usr/bin/env python3
"""bench.py -- MD steps/s of the reactive LJ melt (BASELINE.json metric) on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one velocity-Verlet MD step of the whole system (neighbour-list upkeep, pair and
bonded forces, Langevin thermostat, and the reaction scan every `interval` steps), timed like the
reference's integratorLoop (start_simulation.py:779-781): wall time inside run() only, inputs
already resident in HBM.  Prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--n", type=int, default=1000000, help="particles (C5: 1,000,000 = 100^3 sc lattice)")
    p.add_argument("--rho", type=float, default=0.8)
    p.add_argument("--interval", type=int, default=500)
    p.add_argument("--precision", type=int, default=32)
    p.add_argument("--cpu-steps", type=int, default=8, help="oracle steps for the cpu_baseline leg (0 = skip)")
    p.add_argument("--tpp", type=int, default=0)
    p.add_argument("--opt", action="append", default=[], help="name=value engine option (tuning)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--verbose", action="store_true", help="timers to stderr")
    return p.parse_args()


def cpu_baseline(spec, nsteps):
    """ESPResSo++-algorithm CPU restatement (oracle/, scalar fp64) timed on this host: the timer
    brackets run() only, like the GPU leg.  Bounded sample: `nsteps` steps of the same workload."""
    from chemlab_amd import workloads as W
    from oracle.oracle import OracleEngine
    o = OracleEngine()
    W.apply(spec, o)
    o.run(0)                      # first list build + force evaluation outside the timer (set-up)
    t0 = time.perf_counter()
    o.run(nsteps)
    dt = time.perf_counter() - t0
    reb = o.timers()["rebuilds"] - 1
    o.close()
    return dict(value=nsteps / dt, unit="steps/s", cores=1, kind="port",
                sample="%d steps of the same %d-particle reactive melt (%d list rebuilds inside, no reaction interval), oracle/md_oracle.cpp single thread"
                       % (nsteps, spec["n"], reb))


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if a.gpus == 1 and world == 1:
            pass
        else:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world))
    from chemlab_amd import workloads as W
    from chemlab_amd.engine import Engine

    spec = W.reactive_melt(n=a.n, rho=a.rho, interval=a.interval, seed=2)
    if world > 1 or os.environ.get("CHEM_FORCE_DD"):   # CHEM_FORCE_DD: exercise the RCCL slab path with one rank
        from chemlab_amd import multigpu
        return multigpu.bench_main(a, spec, rank, local_rank, world)

    eng = Engine(device=local_rank, precision=a.precision)
    W.apply(spec, eng)
    if a.tpp:
        eng.set_option("tpp", a.tpp)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, float(v))
    eng.run(a.warmup)
    eng.sync()
    ev0 = len(eng.get_events())
    if not a.no_roofline:
        # dominant per-step kernel = pair force: HIP events on the launch stream around every 8th launch
        # of the TIMED region (the list grows while bonds form, so a sample after the run would be biased)
        eng.set_option("time_pair_kernel", 8)
    t0 = time.perf_counter()
    eng.run(a.steps)
    eng.sync()
    wall = time.perf_counter() - t0
    tm = eng.timers()
    eng.set_option("time_pair_kernel", 0)
    nev = len(eng.get_events()) - ev0
    steps_per_s = a.steps / wall

    out = dict(metric="MD steps/sec, 1M-particle reactive LJ melt", value=steps_per_s, unit="steps/s",
               n_gpus=1, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * wall / a.steps,
               higher_is_better=True, scaling="strong", vs_baseline=None,
               dtype="f32" if a.precision == 32 else "f64", data="synthetic",
               config=dict(workload="C5 reactive LJ melt (chain_growth_catalytic shape): %d particles, rho*=%.4g, rc=2.5, skin=0.3, dt=0.005, Langevin gamma=5 T=0.5, 4 reactions every %d steps"
                                    % (a.n, a.rho, a.interval),
                           particles=a.n, reaction_interval=a.interval, reaction_events=nev,
                           tau_per_day=steps_per_s * spec["dt"] * 86400, parallelism="1 GPU, single domain"))

    if not a.no_roofline:
        launches = tm["pair_kernel_launches"]          # of the last run() = the timed region
        avg_s = 1e-3 * tm["pair_kernel_ms"] / max(launches, 1)
        nb = tm["nlist_entries"] / float(a.n)
        bytes_per_particle = 36.0 + 4.0 * nb      # B_force = 16 (x_i) + 4<nb> + 4 (count) + 16 (f4)
        achieved = a.n * bytes_per_particle / avg_s
        # HBM traffic per launch from the committed PMC passes (rocprofv3 cannot run inside bench.py):
        # 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE, see tools/pmc_traffic.py
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round1_pair_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("particles") == a.n:
                traffic = tj.get("traffic_bytes_per_launch")
        out["roofline"] = dict(bound="hbm", kernel="k_pair_tiles", achieved=achieved / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                               frac=achieved / HBM_PEAK, traffic=traffic, avg_launch_us=avg_s * 1e6, launches=launches,
                               mean_neighbours=nb, algorithmic_bytes_per_particle=bytes_per_particle,
                               whole_step_frac=a.n * (bytes_per_particle + 80.0) * steps_per_s / HBM_PEAK)
    if a.verbose:
        print("timers:", json.dumps(eng.timers()), file=sys.stderr)
    if a.cpu_steps > 0:
        out["cpu_baseline"] = cpu_baseline(spec, a.cpu_steps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
