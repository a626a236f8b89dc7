#!/usr/bin/env python3
"""bench.py -- aggregate MD steps/s of the 1r69 replica ensemble (BASELINE.json configs[1]) on N MI355X.

One bench "step" = one awsem_step launch: `--md-steps` (default 50 = the stage length of the default 2e4-step /
400-stage annealing run, scripts/mm_run.py:98-99 of the reference) Langevin steps of every replica at the
temperature of the current annealing stage (800 -> 200 K).  Inputs (coordinates, velocities, tables) are resident
in HBM before the timed region.  value = replica-MD-steps of all ranks / wall time (max over ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--replicas R] [--md-steps S] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8 TB/s

# workload -> (builder name in openawsem_amd.workloads, builder kwargs, default replicas per GPU, default MD steps per launch, description)
WORKLOADS = {
    "1r69": ("build_1r69", {}, 1024, 50, "1r69 batched annealing ensemble, default forces_setup.py terms, 800->200 K Langevin (dt 2 fs, friction 1/ps)"),
    "1r69-fraglib": ("build_1r69", {"fragment_library": True}, 1024, 50, "1r69 with the 1100-memory fragment library (examples/1r69/frags.mem), default terms, annealing ensemble"),
    "t0958": ("build_t0958", {}, 1024, 50, "T0958 (96 residues), default terms + helical + rama_ssweight, fragment memory, annealing ensemble"),
    "2xov": ("build_2xov", {}, 256, 50, "2xov membrane protein (276 residues), membrane_term + z-dependent contact, annealing ensemble"),
    "2xov-remd": ("build_2xov", {}, 64, 200, "2xov membrane protein, temperature replica exchange 300-600 K, exchange every launch (all-gather of energies, label swap)"),
    "synthetic-hb": ("build_synthetic", {"n_chains": 16, "seed": 3, "hbonds": True}, 1, 100, "synthetic multi-chain system (16 x 2lzm, 2624 residues) with the full backbone + contact + beta 1-3 + P-AP 1-2 + helical force field on the cell-list path"),
    "synthetic-hb-20k": ("build_synthetic", {"n_chains": 122, "hbonds": True}, 1, 20, "synthetic multi-chain system (122 x 2lzm, 20 008 residues) with the full backbone + contact + beta 1-3 + P-AP 1-2 + helical force field on the cell-list path; lambda / gamma tables in functional form (the dense ones would be 5 x 3.2 GB)"),
    "synthetic-60k": ("build_synthetic", {"n_chains": 366}, 1, 20, "synthetic multi-chain system (366 x 2lzm, 60 024 residues), con/chain/chi/excl/rama/contact+burial, cell-list path"),
    "synthetic": ("build_synthetic", {}, 1, 20, "synthetic multi-chain system (122 x 2lzm), con/chain/chi/excl/rama/contact+burial, cell-list path, Langevin (dt 2 fs, friction 1/ps)"),
}


def cpu_baseline(oa, md_steps, budget_s=15.0):
    """The fp64 CPU oracle (kind "port": a restatement, OpenMM itself is not installable here) timed on the host
    cores on a bounded sample of the same workload: `threads` replicas (one per OpenMP thread) of 1r69, same terms,
    same integrator, for as many MD steps as fit ~budget_s."""
    from oracle import awsem_oracle
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count() or 1))
    ctx = awsem_oracle.make_context(oa.system, n_replicas=threads, threads=threads)
    ctx.set_positions(oa.pdb.positions, replica=-1)
    T = np.full(threads, 800.0)
    ctx.set_velocities_to_temperature(T, 2026)
    ctx.step(10, T, 0.002, 1.0, 2026)      # first touch (thread pool, lists)
    t0 = time.perf_counter()
    ctx.step(200, T, 0.002, 1.0, 2026)
    probe = time.perf_counter() - t0
    n = int(max(20, min(400000, budget_s / max(probe / 200, 1e-6))))
    t0 = time.perf_counter()
    ctx.step(n, T, 0.002, 1.0, 2026)
    dt = time.perf_counter() - t0
    return {"value": threads * n / dt, "unit": "replica-MD-steps/s", "cores": threads, "kind": "port",
            "sample": f"{threads} replicas x {n} Langevin steps of 1r69 (default forces_setup.py terms), fp64 C oracle with "
                      f"OpenMP over replicas, {dt:.1f} s; single-trajectory rate {n / dt:.1f} steps/s/core"}


def _measured_traffic(args, oa):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_hbm.json:
    FETCH_SIZE x 2 (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE), when one matches this configuration."""
    import glob
    import re
    best = None
    natural = lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))]   # r01_v9 before r01_v11
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")), key=natural):
        try:
            with open(f) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        if (rec.get("workload") == args.workload and rec.get("replicas") == args.replicas and rec.get("md_steps") == args.md_steps
                and rec.get("precision") == args.precision and bool(rec.get("folded", False)) == bool(args.folded)):
            best = rec
    return None if best is None else best["hbm_bytes_per_launch"]


def _measured_valu(args, folded):
    """VALU instructions per replica-step of the step kernel from the newest committed SQ counter pass of this state
    (profiles/*_pmc_sq.txt: extended chains at 800 K, profiles/*_folded_pmc_sq.txt: compact chains at 300 K; collected by
    tools/pmc_sq.sh on the default workload); (None, None) for any other configuration."""
    import glob
    import re
    if args.workload != "1r69" or args.replicas != 1024 or args.md_steps != 50 or args.precision != "fp32":
        return None, None
    natural = lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))]
    name = re.compile(r"r\d+_v\d+_folded_pmc_sq\.txt$" if folded else r"r\d+_v\d+_pmc_sq\.txt$")   # the 1r69 bench workload only (not *_t0958_*, *_2xov_*)
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_sq.txt")), key=natural) if name.match(os.path.basename(f))]
    if not files:
        return None, None
    rows = {}
    for line in open(files[-1]):
        t = line.split()
        if len(t) >= 2:
            try:
                rows[t[0]] = float(t[1])
            except ValueError:
                pass
    if "SQ_INSTS_VALU" not in rows:
        return None, None
    return rows, os.path.relpath(files[-1], ROOT)


def _valu_view(args, folded, kernel_ms):
    """The issue-rate view of the step kernel: a wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles, a CU has 4 SIMDs
    and holds R / 256 replicas, 2.4 GHz; busy_frac = VALU issue cycles per CU per MD step / CU cycles per MD step."""
    rows, src = _measured_valu(args, folded)
    if rows is None:
        return None
    valu = rows["SQ_INSTS_VALU"]
    per_cu = args.replicas / 256.0
    cu_cycles_per_step = kernel_ms * 1e-3 / (args.md_steps * args.stages) * 2.4e9
    out = {"insts_per_replica_step": valu, "busy_frac": per_cu * valu * 4.0 / 4.0 / cu_cycles_per_step,
           "source": src + " (rocprofv3 --pmc SQ_INSTS_VALU)", "clock_ghz_assumed": 2.4}
    ghz = rows.get("DERIVED_CLOCK_GHZ")
    if ghz:   # the clock the chip actually held under this kernel (GRBM_GUI_ACTIVE of the same counter pass): the 2.4 GHz figure is the boost ceiling
        out["clock_ghz_measured"] = ghz
        out["busy_frac_at_measured_clock"] = per_cu * valu * 4.0 / 4.0 / (kernel_ms * 1e-3 / (args.md_steps * args.stages) * ghz * 1e9)
    return out


def force_eval_leg(ctx, R, n=5):
    """BASELINE.json's second figure: one force evaluation (awsem_forces: pair lists built from scratch, every term, virtual-site
    forces redistributed, fp64 forces written out) of all R replicas at their current coordinates; HIP events around the kernel on the
    engine's stream (awsem_last_kernel_ms), the copy of the 3 x natoms x R forces to the host is outside them.  Mean of n calls after
    one untimed call."""
    ctx.forces()
    ms = []
    for _ in range(n):
        ctx.forces()
        ms.append(ctx.last_kernel_ms())
    k = float(np.mean(ms))
    return {"kernel_ms": k, "us_per_replica": k * 1e3 / R, "calls": n}


def _launch_ranks(n):
    """Start `n` ranks of this script (one process per GPU, as torch.distributed.run would: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*),
    pass their output through (rank 0 prints the JSON line) and return the worst return code.  Called before anything has
    initialised the GPU in this process; the children are fresh interpreters (no fork of a process with GPU state, no exec)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, 16 // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    print(f"[bench] started {len(procs)} ranks (rendezvous 127.0.0.1:{port})", file=sys.stderr, flush=True)
    worst = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.05)
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and worst == 0:
                    worst = rc if rc > 0 else 128 - rc
                    for q in live:               # a rank died: the others would wait in a collective for ever
                        q.terminate()
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        worst = 130
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--replicas", type=int, default=None, help="replicas per GPU (default: 1024 for 1r69, see WORKLOADS)")
    ap.add_argument("--md-steps", type=int, default=None, help="Langevin steps per launch (default 50; 200 for 2xov-remd)")
    ap.add_argument("--stages", type=int, default=1, help="temperature stages of --md-steps steps per launch (awsem_step_stages; default 1 = one awsem_step per stage)")
    ap.add_argument("--precision", choices=["fp32", "fp64"], default="fp32")
    ap.add_argument("--workload", choices=list(WORKLOADS), default="1r69",
                    help="1r69: BASELINE.json configs[1], the configuration the metric is quoted on (default); 1r69-fraglib / t0958: "
                         "configs[2] (many-memory fragment library, all H-bond terms); 2xov / 2xov-remd: configs[3] (membrane protein; "
                         "-remd adds a temperature replica exchange after every launch: all-gather of energies over RCCL, label swap, "
                         "velocity rescale); synthetic: configs[4], one ~20k-residue multi-chain system on the cell-list path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--folded", action="store_true", help="untimed: anneal the ensemble 800 -> 300 K first (400 x 250 steps), then time constant-temperature "
                                                        "launches at 300 K -- the compact state a production run spends most of its time in")
    ap.add_argument("--no-schedule-legs", action="store_true", help="skip the two extra legs of the default workload (whole 400-stage schedule, compact state at 300 K)")
    ap.add_argument("--minimize", action="store_true", help="relax the start structure on the device first (awsem_minimize), as awsem_run does")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: let the N ranks share the GPUs that are there (rank r on device r mod count) with gloo "
                                                             "collectives -- exercises the launcher and the multi-rank code path on a one-GPU box")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: this process becomes the launcher.  It has not touched torch or HIP (and never
        # will), starts one fresh child per rank, relays rank 0's JSON line and exits with the worst return code.
        sys.exit(_launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product path has no CPU fallback)")
    device_index = local_rank
    backend = "nccl"
    if args.share_gpu:
        # rehearsal of the multi-rank path on a box with fewer GPUs than ranks: ranks share devices, and since RCCL refuses two ranks
        # on one device the collectives go through gloo (host memory).  Never a scaling number: the line says "rehearsal".
        device_index = local_rank % torch.cuda.device_count()
        backend = "gloo"
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo")
    coll_device = "cuda" if backend == "nccl" else "cpu"     # where the timing reductions live
    local_rank = device_index

    from openawsem_amd import _capi, workloads
    synthetic = args.workload.startswith("synthetic")
    builder, kwargs, def_replicas, def_steps, description = WORKLOADS[args.workload]
    if args.replicas is None:
        args.replicas = def_replicas
    if args.md_steps is None:
        args.md_steps = def_steps
    oa = getattr(workloads, builder)(**kwargs)
    remd = None
    if args.workload.endswith("-remd"):
        from openawsem_amd import ensemble
        n_total = args.replicas * world
        remd = ensemble.ReplicaExchange(np.linspace(300.0, 600.0, n_total), rank=rank, world=world, seed=2026,
                                        device=torch.device("cuda", local_rank))
    api = _capi.load_hip_library()
    R = args.replicas
    ctx = _capi.Context(api, oa.natoms, n_replicas=R, device=local_rank, precision=0 if args.precision == "fp32" else 1)
    oa.system.upload(ctx)
    ctx.set_positions(oa.pdb.positions, replica=-1)
    if args.minimize and not synthetic:   # as scripts/mm_run.py:153 does before the annealing loop (untimed set-up)
        ctx.minimize(10.0, 0)
    seed = 2026 + 7919 * rank     # independent streams per rank (replica ids repeat across ranks)
    Tstart, Tend, stages = 800.0, 200.0, 400
    ctx.set_velocities_to_temperature(np.full(R, Tstart), seed)

    def stage_T(i):
        return np.full(R, Tstart + (Tend - Tstart) / stages * (i % stages))

    if remd is not None:   # the ladder assignment, temperatures and velocity factors live on the device from here on
        ctx.set_velocities_to_temperature(remd.local_temperatures(), seed)
        remd.attach(ctx)

    def launch(it):
        if remd is not None:
            # MD at the current ladder assignment, then one exchange sweep: energies into a device buffer -> (all-gather over RCCL
            # when the ladder spans ranks) -> sweep kernel -> temperatures and velocity factors, nothing through the host
            ctx.step_remd(args.md_steps, 0.002, 1.0, seed)
            remd.exchange_on_device()
        elif args.stages > 1:   # the same stages, several to a launch
            ctx.step_stages(args.md_steps, np.stack([stage_T(it * args.stages + g) for g in range(args.stages)]), 0.002, 1.0, seed)
        else:
            ctx.step(args.md_steps, stage_T(it), 0.002, 1.0, seed)   # asynchronous: the launches queue up back to back

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def anneal_to_300K():   # untimed: 10^5 steps per replica, 800 -> 300 K, after which the chains are compact
        for i in range(400):
            ctx.step(250, np.full(R, 800.0 - 500.0 * i / 400), 0.002, 1.0, seed)
        ctx.synchronize()

    if args.folded and remd is None:
        anneal_to_300K()
        stage_T = lambda i: np.full(R, 300.0)   # noqa: E731
    it = 0
    for _ in range(args.warmup):
        launch(it)
        it += 1
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        launch(it)
        it += 1
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms.append(ctx.last_kernel_ms())     # mean over the timed launches: HIP events on the engine's stream around each
    if world > 1:
        t = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # replica exchange: what one exchange costs with the launch's lists in place -- back-to-back sweeps on unchanged positions (energy
    # kernel walking the lists the last launch left, group sum, [all-gather,] sweep kernel; the velocity factors accumulate on the device for
    # the next step launch to take in), outside the timed region
    exchange_us = None
    if remd is not None:
        barrier()
        remd.exchange_on_device()      # a replica that ended the launch beyond skin/2 rebuilds its lists in this one (the next launch's rebuild)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            remd.exchange_on_device()
        barrier()
        exchange_us = (time.perf_counter() - t0) / args.steps * 1e6
        if world > 1:
            t = torch.tensor([exchange_us], device=coll_device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            exchange_us = float(t.item())

    # sanity: the trajectories are alive (finite energies), not part of the timed region
    e = ctx.energy()
    if not np.all(np.isfinite(e)):
        sys.exit("non-finite potential energy after the benchmark run")

    # force-evaluation cost on the frames the timed launches ended at (the headline's extended chains, ~790 K; compact chains with --folded)
    force_eval = {}
    if remd is None and not synthetic:
        force_eval["compact_300K" if args.folded else "extended_800K"] = force_eval_leg(ctx, R)

    def all_ranks_max(t):
        if world > 1:
            tt = torch.tensor([t], device=coll_device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return t

    # Two more legs on the headline configuration (the default workload only): the headline times the first stages of the
    # schedule (extended chains, ~790 K), where a replica has the fewest pairs inside its cutoffs; a production run spends most of
    # its time elsewhere.  (1) the WHOLE default schedule of scripts/mm_run.py:98,169-175 of the reference -- 400 stages x 50
    # steps, 800 -> 200 K -- from a fresh start; (2) constant 300 K after an untimed 10^5-step anneal (compact chains).
    legs = {}
    if args.workload == "1r69" and remd is None and not args.folded and not args.no_schedule_legs:
        ctx.set_positions(oa.pdb.positions, replica=-1)
        ctx.set_velocities_to_temperature(np.full(R, Tstart), seed)
        barrier()
        t0 = time.perf_counter()
        for i in range(stages):
            ctx.step(args.md_steps, np.full(R, Tstart + (Tend - Tstart) / stages * i), 0.002, 1.0, seed)
        barrier()
        t_full = all_ranks_max(time.perf_counter() - t0)
        legs["full_schedule"] = {"value": world * R * args.md_steps * stages / t_full, "seconds": t_full, "kernel_ms": ctx.last_kernel_ms(),
                                 "what": f"{stages} stages x {args.md_steps} steps, {Tstart:.0f} -> {Tend:.0f} K, fresh extended start: the default awsem_run schedule"}
        # the same schedule four stages to a launch (awsem_step_stages; same trajectories bit for bit): what engine.Context sends when the
        # run reports less often than it changes the temperature (awsem_run --reportFrequency 200 and up; at the default, a report after
        # every stage, each stage stays a launch of its own and the leg above is the one that applies)
        STAGED = 4
        ctx.set_positions(oa.pdb.positions, replica=-1)
        ctx.set_velocities_to_temperature(np.full(R, Tstart), seed)
        barrier()
        t0 = time.perf_counter()
        for i in range(0, stages, STAGED):
            ctx.step_stages(args.md_steps, np.stack([np.full(R, Tstart + (Tend - Tstart) / stages * (i + g)) for g in range(STAGED)]), 0.002, 1.0, seed)
        barrier()
        t_fs = all_ranks_max(time.perf_counter() - t0)
        legs["full_schedule_staged"] = {"value": world * R * args.md_steps * stages / t_fs, "seconds": t_fs, "kernel_ms": ctx.last_kernel_ms(),
                                        "what": f"the same {stages} stages, {STAGED} to a launch (awsem_step_stages): runs that report every {STAGED} stages or less often"}
        ctx.set_positions(oa.pdb.positions, replica=-1)
        ctx.set_velocities_to_temperature(np.full(R, Tstart), seed)
        anneal_to_300K()
        for _ in range(args.warmup):
            ctx.step(args.md_steps, np.full(R, 300.0), 0.002, 1.0, seed)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.step(args.md_steps, np.full(R, 300.0), 0.002, 1.0, seed)
        barrier()
        t_c = all_ranks_max(time.perf_counter() - t0)
        legs["compact_300K"] = {"value": world * R * args.md_steps * args.steps / t_c, "seconds": t_c, "kernel_ms": ctx.last_kernel_ms(),
                                "what": f"{args.steps} launches x {args.md_steps} steps at 300 K after an untimed 400 x 250-step anneal 800 -> 300 K (compact chains)"}
        barrier()
        t0 = time.perf_counter()
        for _ in range(max(1, args.steps // STAGED)):
            ctx.step_stages(args.md_steps, np.full((STAGED, R), 300.0), 0.002, 1.0, seed)
        barrier()
        t_cs = all_ranks_max(time.perf_counter() - t0)
        legs["compact_300K_staged"] = {"value": world * R * args.md_steps * STAGED * max(1, args.steps // STAGED) / t_cs, "seconds": t_cs, "kernel_ms": ctx.last_kernel_ms(),
                                       "what": f"{max(1, args.steps // STAGED)} launches x {STAGED} stages x {args.md_steps} steps at 300 K, continuing the leg above"}
        if not np.all(np.isfinite(ctx.energy())):
            sys.exit("non-finite potential energy after the schedule legs")
        force_eval["compact_300K"] = force_eval_leg(ctx, R)

    if rank == 0:
        total_steps = world * R * args.md_steps * args.steps * args.stages
        b_alg, n_real, n_fm = workloads.algorithmic_bytes_per_replica_step(oa)
        if synthetic:   # ~110 B per site and step on the neighbour-list path
            b_alg = 110 * oa.nres
        k_ms = float(np.mean(kernel_ms))
        achieved = b_alg * R * args.md_steps * args.stages / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "MD steps/sec (all replicas)", "value": total_steps / elapsed, "unit": "replica-MD-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f64",
            "data": (f"synthetic ({kwargs.get('n_chains', 122)} rigidly rotated copies of the reference's 2lzm test structure on a jittered lattice, seed {kwargs.get('seed', 2026)})" if synthetic else
                     f"synthetic ({args.workload.split('-')[0]} start structure from the reference's examples/, Maxwell-Boltzmann velocities, Philox seed 2026)"),
            "config": {"workload": description + (" -- timed at 300 K after an untimed anneal (compact chains)" if args.folded else ""), "replicas_per_gpu": R, "md_steps_per_launch": args.md_steps * args.stages, "stages_per_launch": args.stages,
                       "residues": oa.nres, "atoms": oa.natoms, "parallelism": f"replica-sharded x{world}" + (" (REHEARSAL: ranks share GPUs, gloo collectives)" if args.share_gpu and world > 1 else "")},
            # the whole MD step (forces + virtual sites + list upkeep + Langevin update) per replica, from the step kernel's own time
            "md_step_us_per_replica_step": k_ms * 1e3 / (R * args.md_steps * args.stages),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": _measured_traffic(args, oa), "kernel": "cell-list step (kernel sequence)" if synthetic else "awsem_step_kernel<float>", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_replica_step": b_alg,
                         "note": ("cell-list path: a step is a sequence of five grid-wide kernels (seven with hydrogen-bond terms; graph replay), each "
                                  "about one round of waves long at this size (see DESIGN.md sections 8 and 11.1)") if synthetic else
                                 ("state is LDS-resident across the launch, so the kernel is VALU/LDS-bound, not HBM-bound "
                                  "(see DESIGN.md); the HBM fraction is reported because the metric asks for it")},
        }
        if force_eval:
            # BASELINE.json's "force-eval us/step": a standalone awsem_forces call (lists from scratch + all terms + fp64 output), per replica
            first = "extended_800K" if "extended_800K" in force_eval else "compact_300K"
            out["force_eval_us_per_replica_step"] = force_eval[first]["us_per_replica"]
            out["force_eval"] = force_eval
        valu = None if synthetic else _valu_view(args, args.folded, k_ms)
        if valu is not None:
            # The state is LDS-resident, so the roof this kernel actually sits under is VALU issue, not HBM: that fraction is the
            # primary figure; `frac` (HBM, by SURVEY section 8(d)'s streaming formula) stays because the metric asks for it.
            out["roofline"]["primary_bound"] = "valu_issue"
            out["roofline"]["valu"] = valu
        if legs:
            # the whole-schedule truth next to the headline: the driver-parsed line carries all three rates
            out["value_full_schedule"] = legs["full_schedule"]["value"]
            out["value_compact_300K"] = legs["compact_300K"]["value"]
            out["value_full_schedule_staged"] = legs["full_schedule_staged"]["value"]
            out["value_compact_300K_staged"] = legs["compact_300K_staged"]["value"]
            out["legs"] = legs
            vc = _valu_view(args, True, legs["compact_300K"]["kernel_ms"])
            if vc is not None:
                out["roofline"]["valu_compact_300K"] = vc
        if remd is not None:
            remd.pull()
            out["config"]["remd_acceptance"] = [round(float(a), 3) for a in remd.acceptance()[:8]]
            # What one exchange costs once the lists are in place (measured above, back to back), and the in-situ figure: wall time per
            # iteration minus the step kernel's own time.  The latter also holds the list rebuilds of the replicas that ended the launch
            # more than skin/2 from where their lists were built: they rebuild in the energy kernel and hand the lists to the next
            # launch, which would otherwise have begun with that rebuild.
            out["remd_exchange_us"] = exchange_us
            out["remd_iteration_minus_step_kernel_us"] = (elapsed / args.steps - k_ms * 1e-3) * 1e6
            out["config"]["remd_exchange"] = ("per iteration: energy kernel + group sum into a device buffer, "
                                              + ("one all_gather_into_tensor over RCCL, " if world > 1 else "") + "sweep kernel (Metropolis, ladder index swap); the velocity rescale rides in the next step launch's velocity load; no host copy")
        if not args.no_cpu_baseline and world == 1 and args.workload == "1r69":
            out["cpu_baseline"] = cpu_baseline(oa, args.md_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
