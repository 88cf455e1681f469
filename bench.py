#!/usr/bin/env python3
"""bench.py -- MD steps/s of the reactive LJ melt (BASELINE.json metric) on N MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts the N ranks itself, as a child torch.distributed.run)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one velocity-Verlet MD step of the whole system (neighbour-list upkeep, pair and
bonded forces, Langevin thermostat, and the reaction scan every `interval` steps), timed like the
reference's integratorLoop (start_simulation.py:779-781): wall time inside run() only, inputs
already resident in HBM.  Prints ONE JSON line (see DESIGN.md "Measurement").

Sequence (all set-up outside the timer):
  1. lattice start, `--equil` steps WITHOUT reactions: the melt the metric is named after
     (the reference attaches the reaction extension at step `start_ar`, start_simulation.py:735-741);
  2. reactions enabled, W warm-up steps, then EXACTLY K timed steps;
  3. if no reaction step fell into the timed region (K < interval), the next reaction step is timed
     on its own afterwards and reported as `reaction_step_ms` (it is never folded into `value`);
  3b. `late_stage`: 10 more reaction steps (untimed), then max(K, 400) timed steps of the gelled system;
  4. an fp64 run of the same melted state (`f64` object: the reference computes in fp64);
  5. the CPU restatement on the same melted state, one core and all cores (`cpu_baseline`).
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
    # (`--particles`, not `--n`: torch.distributed.run's own parser claims every `--n...` prefix as ambiguous)
    p.add_argument("--particles", "--n", dest="n", type=int, default=1000000, help="particles (C5: 1,000,000 = 100^3 sc lattice)")
    p.add_argument("--rho", type=float, default=0.8)
    p.add_argument("--skin", type=float, default=0.3, help="Verlet skin of the workload (BASELINE C5: 0.3)")
    p.add_argument("--interval", type=int, default=500)
    p.add_argument("--equil", type=int, default=2000, help="untimed melting steps before reactions are enabled")
    p.add_argument("--precision", type=int, default=32)
    p.add_argument("--f64-steps", type=int, default=-1, help="timed steps of the fp64 leg (-1: min(K, 400); 0 = skip)")
    p.add_argument("--cpu-steps", type=int, default=-1, help="oracle steps per cpu_baseline leg (-1: automatic, 0 = skip)")
    p.add_argument("--tpp", type=int, default=0)
    p.add_argument("--opt", action="append", default=[], help="name=value engine option (tuning)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--late-stage", type=int, default=10, help="reaction steps to advance (untimed) before the late-stage leg (0 = skip)")
    p.add_argument("--verbose", action="store_true", help="timers to stderr")
    p.add_argument("--rendezvous-only", action="store_true",
                   help="N > 1: start the ranks, run the gloo rendezvous (barrier + max over ranks) and stop before any GPU work")
    return p.parse_args()


WORKLOAD = ("C5 reactive LJ melt (chain_growth_catalytic shape): %d particles, rho*=%.4g, rc=2.5, skin=%.3g, dt=0.005, "
            "Langevin gamma=5 T=0.5, 4 reactions every %d steps; melted for %d steps before the reactions start")


def melted_spec(spec, eng):
    """The same workload with the engine's current (melted) positions and velocities: what the fp64 leg and
    the CPU baseline start from.  Types/states are still the initial ones (reactions were off so far)."""
    s = dict(spec)
    s["pos"] = eng.get_state("POS")
    s["vel"] = eng.get_state("VEL")
    return s


def cpu_baseline(spec, nsteps, cores_all):
    """ESPResSo++-algorithm CPU restatement (oracle/, fp64) timed on this host's cores: the timer brackets
    run() only, like the GPU leg.  Bounded sample of the same workload; -O3 -march=native -fopenmp build
    (oracle/Makefile: liboracle_omp.so), once with one thread and once with all cores."""
    from chemlab_amd import workloads as W
    from oracle.oracle import OracleEngine
    res = {}
    for label, th, ns in (("one_core", 1, max(2, nsteps // 4)), ("all_cores", cores_all, nsteps)):
        o = OracleEngine(threads=th)
        W.apply(spec, o)
        o.run(0)                      # first list build + force evaluation outside the timer (set-up)
        r0 = o.timers()["rebuilds"]
        t0 = time.perf_counter()
        o.run(ns)
        dt = time.perf_counter() - t0
        res[label] = dict(value=ns / dt, steps=ns, rebuilds=o.timers()["rebuilds"] - r0, cores=th)
        o.close()
    a, b = res["all_cores"], res["one_core"]
    return dict(value=a["value"], unit="steps/s", cores=a["cores"], kind="port",
                sample="%d steps (%d list rebuilds inside, no reaction interval) of the same melted %d-particle workload, "
                       "oracle/md_oracle.cpp built -O3 -march=native -fopenmp, OpenMP over cells; one core: %d steps (%d rebuilds)"
                       % (a["steps"], a["rebuilds"], spec["n"], b["steps"], b["rebuilds"]),
                one_core=dict(value=b["value"], unit="steps/s", cores=1))


def kernel_rooflines(a, tm, steps_per_s, precision):
    """HBM-roofline figures of the per-step kernels from the HIP-event samples of the timed region.
    Algorithmic bytes per particle, restated for this build's layout as 8d demands:
      pair force  B = 16 (x_i) + w<nb> + 4 (count) + 16 (f4) with w = 2 bytes per entry (16-bit tile-local slots)
                  and <nb> = entries of the force list actually read (pairs that carry a potential);
                  SURVEY's int32 formula (w = 4) is given beside it;
      rebuild     B = 16+16 (sort read/write x4) + w<nb> (list write) + 8;
      integrate   B = 80 (read x4,v4,f4; write x4,v4), x2 in fp64."""
    n = float(a.n)
    rb = 2.0 if precision == 64 else 1.0      # fp64 parity mode doubles the x/v/f terms
    nb = tm["nlist_entries"] / n
    out = []

    def entry(name, ms, launches, bpp, extra):
        if launches <= 0:
            return None
        avg_s = 1e-3 * ms / launches
        ach = n * bpp / avg_s
        d = dict(kernel=name, bound="hbm", achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=ach / HBM_PEAK,
                 avg_launch_us=avg_s * 1e6, launches_sampled=int(launches), algorithmic_bytes_per_particle=bpp)
        d.update(extra)
        return d

    bpp_pair = 16.0 * rb + 2.0 * nb + 4.0 + 16.0 * rb
    e = entry("k_pair_tiles", tm["pair_kernel_ms"], tm["pair_kernel_launches"], bpp_pair,
              dict(mean_neighbours=nb, survey_int32_bytes_per_particle=36.0 + 4.0 * nb))
    if e:
        e["frac_survey_int32_formula"] = n * (36.0 + 4.0 * nb) / (e["avg_launch_us"] * 1e-6) / HBM_PEAK
        out.append(e)
    bpp_reb = 32.0 * rb + 2.0 * nb + 8.0
    e = entry("k_rebuild_fused (rebuilding launches)", tm["rebuild_kernel_ms"], tm["rebuild_kernel_launches"], bpp_reb,
              dict(survey_int32_bytes_per_particle=40.0 + 4.0 * 73.6))
    if e:
        e["frac_survey_int32_formula"] = n * (40.0 + 4.0 * 73.6) / (e["avg_launch_us"] * 1e-6) / HBM_PEAK
        out.append(e)
    e = entry("k_integrate", tm["integrate_kernel_ms"], tm["integrate_kernel_launches"], 80.0 * rb, {})
    if e:
        out.append(e)
    return out


def time_share(tm, steps):
    """Per-step device time of each kernel class (us), from the sampled launches: the neighbour kernel runs
    every step (decision only on most, rebuild on some)."""
    def avg(ms, k):
        return 1e3 * ms / k if k > 0 else 0.0
    nsamp = tm["rebuild_kernel_launches"] + tm["decide_kernel_launches"]
    share = dict(pair=avg(tm["pair_kernel_ms"], tm["pair_kernel_launches"]),
                 integrate=avg(tm["integrate_kernel_ms"], tm["integrate_kernel_launches"]),
                 bonded=avg(tm["bonded_kernel_ms"], tm["bonded_kernel_launches"]),
                 neighbour=1e3 * (tm["rebuild_kernel_ms"] + tm["decide_kernel_ms"]) / nsamp if nsamp else 0.0)
    return share


def load_traffic(n):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 cannot wrap itself inside bench.py):
    2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, tools/pmc_traffic.py."""
    for name in ("round3_kernel_traffic.json", "round2_kernel_traffic.json", "round1_pair_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                tj = json.load(f)
            if tj.get("particles") == n:
                return tj, "profiles/" + name
    return None, None


def run_leg(a, spec, precision, local_rank, equil_state=None, steps=None, warmup=None):
    """One engine through equilibration (or a given melted state), warm-up and the timed region."""
    from chemlab_amd import workloads as W
    from chemlab_amd.engine import Engine
    steps = a.steps if steps is None else steps
    warmup = a.warmup if warmup is None else warmup
    eng = Engine(device=local_rank, precision=precision)
    W.apply(spec if equil_state is None else equil_state, eng)
    if a.tpp:
        eng.set_option("tpp", a.tpp)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, float(v))
    info = {}
    if equil_state is None:
        eng.reactions_enable(False)
        x0 = np.asarray(spec["pos"], dtype=np.float64)
        eng.run(a.equil)
        eng.sync()
        if a.equil > 0:
            d = eng.get_state("POS_UNFOLDED") - x0
            info["msd_after_melting"] = float((d * d).sum(1).mean())   # lattice spacing^2 = 1.16: > that means the lattice is gone
        info["melted"] = melted_spec(spec, eng)
        eng.reactions_enable(True)
    eng.run(warmup)
    eng.sync()
    ev0 = len(eng.get_events())
    tm0 = eng.timers()
    if not a.no_roofline:
        # HIP events on the launch stream around the per-step kernels of every N-th step of the TIMED region
        # (the event records between the launches cost ~20 us on a sampled step -- with every step of a 20-step region
        #  sampled, `value` came out 18 % low; every 5th: 2.8 % low (7097 against 7297 steps/s).  Short regions are sampled
        #  every 10th step, long ones every 5th at least)
        eng.set_option("time_pair_kernel", max(10 if steps < 200 else 5, steps // 256))
    t0 = time.perf_counter()
    eng.run(steps)
    eng.sync()
    wall = time.perf_counter() - t0
    tm = eng.timers()
    eng.set_option("time_pair_kernel", 0)
    info.update(wall=wall, tm=tm, events=len(eng.get_events()) - ev0, reaction_steps_timed=tm["reaction_steps"] - tm0["reaction_steps"],
                rebuilds_timed=tm["rebuilds"] - tm0["rebuilds"], list_builds_timed=tm["list_rebuilds"] - tm0["list_rebuilds"],
                reaction_wall_s=tm["reaction_wall_s"] - tm0["reaction_wall_s"])
    return eng, info


def time_reaction_step(eng, interval, ms_per_step):
    """Advance (untimed) to the step before the next reaction step, then run that one step alone.
    `reaction_step_ms`: the engine's own timer around the reaction step (scan, resolve, apply, host topology update) -- the same
    definition as when reaction steps fall inside the timed region (reaction_wall_s / count), so that `steady_state_steps_per_s`
    predicts what a long run measures (2000 steps with 4 reaction steps inside: 7.5k steps/s = 500 x 0.1265 ms + 2.9 ms).
    `reaction_step_call_ms`: wall time of the whole one-step call minus one MD step -- it also holds what a long run hides behind
    the following steps (the bookkeeping thread joined at the end of the call, the call's own opening force evaluation)."""
    togo = (interval - 1) - (eng.step % interval)
    if togo < 0:
        togo += interval
    eng.run(togo)
    eng.sync()
    ev0 = len(eng.get_events())
    tm0 = eng.timers()
    t0 = time.perf_counter()
    eng.run(1)
    eng.sync()
    dt = time.perf_counter() - t0
    tm1 = eng.timers()
    return dict(reaction_step_ms=1e3 * (tm1["reaction_wall_s"] - tm0["reaction_wall_s"]), reaction_step_call_ms=max(0.0, 1e3 * dt - ms_per_step),
                reaction_events=len(eng.get_events()) - ev0, at_step=int(eng.step))


def spawn_ranks(nranks):
    """`python bench.py --gpus N` started plainly (no torch.distributed.run around it): this process never touches the
    GPU -- it starts the N ranks as a CHILD process (`python -m torch.distributed.run ... bench.py <same arguments>`, one
    rank per GPU, rendezvous on 127.0.0.1), relays the child's output (rank 0's JSON line) and exits with its code."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, env=env)
    return res.returncode


def late_stage_leg(eng, a, n_reaction_steps=10):
    """The gelled regime reactive runs spend most of their life in: advance (untimed) through `n_reaction_steps` more
    reaction steps, then time max(K, 400) steps that start right behind a reaction step (so that none falls inside) with
    the same per-kernel HIP-event sampling as the headline region."""
    togo = n_reaction_steps * a.interval - (eng.step % a.interval)
    eng.run(togo)
    eng.sync()
    k = min(max(a.steps, 400), a.interval - 1)
    tm0 = eng.timers()
    eng.set_option("time_pair_kernel", max(5, k // 256))
    t0 = time.perf_counter()
    eng.run(k)
    eng.sync()
    wall = time.perf_counter() - t0
    tm = eng.timers()
    eng.set_option("time_pair_kernel", 0)
    obs = eng.observe()
    bonds = int(sum(obs["list_size"]))
    share = time_share(tm, k)
    ks = kernel_rooflines(a, tm, k / wall, a.precision)
    return dict(value=k / wall, unit="steps/s", steps=k, ms_per_step=1e3 * wall / k, at_step=int(eng.step), reaction_steps_before=int(tm0["reaction_steps"]),
                bonds=bonds, conversion=bonds / (0.5 * a.n), events=int(len(eng.get_events())), list_rebuilds_timed=int(tm["rebuilds"] - tm0["rebuilds"]), list_builds_timed=int(tm["list_rebuilds"] - tm0["list_rebuilds"]),
                device_us_per_step=dict(pair=share["pair"], neighbour_kernel=share["neighbour"], integrate=share["integrate"], bonded=share["bonded"]),
                kernels=[dict(kernel=q["kernel"], avg_launch_us=q["avg_launch_us"], frac=q["frac"], launches_sampled=q["launches_sampled"]) for q in ks])


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or plainly "
                         "(python bench.py --gpus %d starts the ranks itself)" % (a.gpus, world, a.gpus, a.gpus))
    from chemlab_amd import workloads as W

    if a.rendezvous_only:
        from chemlab_amd import multigpu
        dist = multigpu.init_process_group()
        dist.barrier()
        top = multigpu.max_over_ranks(dist, float(rank))
        if rank == 0:
            print(json.dumps(dict(rendezvous="ok", n_gpus=world, max_rank=int(top), node_grid=list(multigpu.node_grid(world)))))
        dist.barrier()
        return 0
    spec = W.reactive_melt(n=a.n, rho=a.rho, interval=a.interval, seed=2, skin=a.skin)
    if world > 1 or os.environ.get("CHEM_FORCE_DD"):   # CHEM_FORCE_DD: exercise the RCCL slab path with one rank
        from chemlab_amd import multigpu
        return multigpu.bench_main(a, spec, rank, local_rank, world)

    eng, info = run_leg(a, spec, a.precision, local_rank)
    wall, tm = info["wall"], info["tm"]
    steps_per_s = a.steps / wall
    ms_per_step = 1e3 * wall / a.steps

    out = dict(metric="MD steps/sec, 1M-particle reactive LJ melt", value=steps_per_s, unit="steps/s",
               n_gpus=1, steps=a.steps, warmup=a.warmup, ms_per_step=ms_per_step,
               higher_is_better=True, scaling="strong", vs_baseline=None,
               dtype="f32" if a.precision == 32 else "f64", data="synthetic",
               config=dict(workload=WORKLOAD % (a.n, a.rho, a.skin, a.interval, a.equil),
                           particles=a.n, reaction_interval=a.interval, equilibration_steps=a.equil,
                           msd_after_melting=info.get("msd_after_melting"),
                           reaction_steps_timed=int(info["reaction_steps_timed"]), reaction_events=int(info["events"]),
                           list_rebuilds_timed=int(info["rebuilds_timed"]), list_builds_timed=int(info["list_builds_timed"]),
                           tau_per_day=steps_per_s * spec["dt"] * 86400, parallelism="1 GPU, single domain"))
    if info["reaction_steps_timed"] > 0:
        out["config"]["reaction_step_ms"] = 1e3 * info["reaction_wall_s"] / info["reaction_steps_timed"]
    else:
        # K < interval: the reaction step the metric's name promises is measured on its own, right after the
        # timed region, and reported beside `value` (never folded into it)
        r = time_reaction_step(eng, a.interval, ms_per_step)
        out["config"].update(reaction_step_ms=r["reaction_step_ms"], reaction_step_call_ms=r["reaction_step_call_ms"], reaction_events=int(r["reaction_events"]),
                             reaction_step_measured_separately_at_step=r["at_step"],
                             steady_state_steps_per_s=a.interval / (a.interval * ms_per_step * 1e-3 + r["reaction_step_ms"] * 1e-3))

    if not a.no_roofline:
        ks = kernel_rooflines(a, tm, steps_per_s, a.precision)
        share = time_share(tm, a.steps)
        traffic, src = load_traffic(a.n)
        for k in ks:
            key = "k_pair_tiles" if k["kernel"].startswith("k_pair") else ("k_rebuild_fused" if k["kernel"].startswith("k_rebuild") else "k_integrate")
            k["traffic"] = (traffic or {}).get("traffic_bytes_per_launch", {}).get(key) if isinstance((traffic or {}).get("traffic_bytes_per_launch"), dict) \
                else ((traffic or {}).get("traffic_bytes_per_launch") if key == "k_pair_tiles" else None)
            k["traffic_source"] = src if k["traffic"] is not None else None
        # the dominant kernel = the one with the largest share of the device time per step
        per_step = dict(k_pair_tiles=share["pair"], k_rebuild_fused=share["neighbour"], k_integrate=share["integrate"])
        dom = max(per_step, key=per_step.get)
        main_k = next((k for k in ks if k["kernel"].startswith(dom)), ks[0] if ks else None)
        if main_k:
            rl = dict(main_k)
            rl["device_us_per_step"] = dict(pair=share["pair"], neighbour_kernel=share["neighbour"], integrate=share["integrate"], bonded=share["bonded"])
            rl["kernels"] = ks
            nb = tm["nlist_entries"] / float(a.n)
            rl["whole_step_frac"] = a.n * (36.0 + 2.0 * nb + 80.0) * steps_per_s / HBM_PEAK
            out["roofline"] = rl
    if a.late_stage > 0 and not a.no_roofline:
        # `conversion` = chain bonds per A bead (n/2 of them); the headline region above has none yet
        out["late_stage"] = late_stage_leg(eng, a, a.late_stage)
    if a.verbose:
        print("timers:", json.dumps(tm), file=sys.stderr)
    melted = info["melted"]
    eng.close()

    f64_steps = a.f64_steps if a.f64_steps >= 0 else min(a.steps, 400)
    if f64_steps > 0 and a.precision == 32:
        # the reference computes in fp64: the same melted state through the fp64 build of every kernel
        e64, i64 = run_leg(a, spec, 64, local_rank, equil_state=melted, steps=f64_steps, warmup=min(a.warmup, 50))
        sps64 = f64_steps / i64["wall"]
        o64 = dict(value=sps64, unit="steps/s", dtype="f64", steps=f64_steps, ms_per_step=1e3 * i64["wall"] / f64_steps,
                   reaction_steps_timed=int(i64["reaction_steps_timed"]), list_rebuilds_timed=int(i64["rebuilds_timed"]))
        if not a.no_roofline:
            o64["kernels"] = kernel_rooflines(a, i64["tm"], sps64, 64)
        out["f64"] = o64
        e64.close()

    if a.cpu_steps != 0:
        from oracle.oracle import host_cores, lscpu_cores
        # a one-GPU box exposes every core of the host in the affinity mask but grants a 16-core share
        # (more threads than that only fight each other: 256 threads ran 3.5x SLOWER than one)
        cores = min(host_cores(), 16 * max(1, a.gpus))
        ns = a.cpu_steps if a.cpu_steps > 0 else max(8, int(round(12 * 1.0e6 / a.n * min(cores, 16) / 8.0)))
        out["cpu_baseline"] = cpu_baseline(melted, ns, cores)
        lc = lscpu_cores()
        out["cpu_baseline"]["host"] = dict(lscpu_logical_cpus=lc[0] if lc else None, lscpu_physical_cores=lc[1] if lc else None,
                                           usable_cores=host_cores(), threads_used=cores)
        out["cpu_baseline"]["sample"] += ("; host per lscpu: %s logical CPUs / %s physical cores, share granted to this job (affinity + cgroup "
                                          "quota): %d, threads used: %d" % (lc[0] if lc else "?", lc[1] if lc else "?", host_cores(), cores))
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main() or 0)
