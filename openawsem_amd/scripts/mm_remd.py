#!/usr/bin/env python3
"""Temperature replica exchange on the MI355X engine: the command line of openawsem/scripts/mm_remd.py (flags :171-190,
set-up :31-105, ladder `np.linspace(tempStart, tempEnd, replicas)` :87, `exchange_frequency` Langevin steps between swap
attempts :92-93,133-141, minimisation of the start structure :108-113).

The reference hands the sampling to openmmtools' ReplicaExchangeSampler in one process and stores a NetCDF file
(output.nc, MultiStateReporter); neither library is part of this build.  Here all replicas of a rank live in one engine
handle (one launch advances them all), an exchange step is one gather of the potential energies, identical Metropolis
decisions on every rank and a swap of temperature labels with a velocity rescale (openawsem_amd.ensemble); the mixing scheme is
openmmtools' default, swap-all, as in the reference (which passes no replica_mixing_scheme, :133-140).  Launch with
`python -m torch.distributed.run --nproc-per-node N ...` to spread the ladder over N GPUs.  Outputs in --to:
  output.npz          temperatures, per-iteration potential energies [iteration][replica] (kJ/mol), ladder index of every
                      replica after each exchange, acceptance per neighbouring pair          (rank 0)
  replicas_rank<r>.npz  positions of this rank's replicas every --reportFrequency iterations, final positions / velocities,
                      ladder state: also the checkpoint that --fromCheckPoint <folder> restarts from
  time.dat            wall time in seconds (rank 0)
"""
import argparse
import os
import shutil
import sys
import time

import numpy as np

import openawsem_amd
from openawsem_amd import engine, ensemble
from openawsem_amd.scripts.mm_run import load_forces_module
from openawsem_amd.unit import femtoseconds, kelvin, picosecond


def run_replica_exchange(args):
    rank, world, local_rank = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    device = None
    if world > 1:
        import torch
        import torch.distributed as dist
        on_gpu = torch.cuda.is_available() and args.platform != "Reference-gloo"
        if not dist.is_initialized():
            dist.init_process_group("nccl" if on_gpu else "gloo")
        device = torch.device("cuda", local_rank) if on_gpu else None

    platform = engine.Platform.getPlatformByName(args.platform)
    platform.setPropertyDefaultValue("DeviceIndex", str(local_rank if args.device is None else args.device))

    setupFolderPath = os.path.dirname(args.protein) or "."
    pdb_id = os.path.basename(args.protein)
    toPath = os.path.abspath(args.to)
    checkPointPath = None if args.fromCheckPoint is None else os.path.abspath(args.fromCheckPoint)
    forceSetupFile = os.path.abspath(args.forces) if args.forces else \
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "forces_setup.py")
    parametersLocation = "." if args.parameters is None else os.path.abspath(args.parameters)
    os.chdir(setupFolderPath)

    chain = args.chain
    if chain == "-1":
        chain = openawsem_amd.getAllChains("crystal_structure.pdb") if os.path.exists("crystal_structure.pdb") else "A"
        print("Chains to simulate: ", chain)
    if rank == 0:
        os.makedirs(toPath, exist_ok=True)
        if args.to != "./":
            shutil.copy(forceSetupFile, os.path.join(toPath, "forces_setup.py"))
            for f in ("crystal_structure.fasta", "crystal_structure.pdb"):
                if os.path.exists(f):
                    shutil.copy(f, toPath)

    suffix = "-openmmawsem.pdb"
    input_pdb_filename = pdb_id if (args.fromOpenMMPDB or pdb_id.endswith(suffix)) else f"{pdb_id}{suffix}"
    seq = None if args.fasta == "" else openawsem_amd.read_fasta(args.fasta)

    # ladder and schedule (mm_remd.py:85-93)
    temps = np.linspace(args.tempStart, args.tempEnd, args.replicas)
    simulation_steps = int(np.floor(args.runtime * 1e6 / args.timeStep))          # ns -> fs / (fs per step)
    exchange_attempts = int(np.floor(simulation_steps / args.exchange_frequency))
    if args.iterations is not None:
        exchange_attempts = args.iterations

    oa = openawsem_amd.OpenMMAWSEMSystem(input_pdb_filename, k_awsem=1.0, chains=chain, xml_filename=openawsem_amd.xml,
                                          seqFromPdb=seq, includeLigands=args.includeLigands)
    forces = load_forces_module(forceSetupFile).set_up_forces(oa, submode=args.subMode, contactParameterLocation=parametersLocation)
    oa.addForcesWithDefaultForceGroup(forces)

    remd = ensemble.ReplicaExchange(temps, rank=rank, world=world, seed=args.seed, device=device, mixing=args.mixing)
    R = len(remd.local)
    integrator = engine.LangevinIntegrator(args.tempStart * kelvin, 1 / picosecond, args.timeStep * femtoseconds)
    integrator.setRandomNumberSeed(args.seed + 7919 * rank)
    simulation = engine.Simulation(oa.pdb.topology, oa.system, integrator, platform, n_replicas=R)
    ctx = simulation.context
    my_file = os.path.join(toPath, f"replicas_rank{rank}.npz")
    if checkPointPath:
        with np.load(os.path.join(checkPointPath, f"replicas_rank{rank}.npz"), allow_pickle=False) as z:
            if z["positions"].shape[0] != R or len(z["slot"]) != remd.n:
                raise ValueError("checkpoint does not match this ladder / rank layout")
            ctx._ctx.set_state(z["positions"], z["velocities"])
            remd.slot, remd.iteration = z["slot"].copy(), int(z["iteration"])
            remd.attempted, remd.accepted = z["attempted"].copy(), z["accepted"].copy()
            ctx._ctx.step_count = int(z["step_count"])
    else:
        ctx.setPositions(oa.pdb.positions)
        simulation.minimizeEnergy()               # every replica starts from the minimised structure (mm_remd.py:108-113)
        ctx.setVelocitiesToTemperature(remd.local_temperatures(), args.seed + 7919 * rank)

    energies, slots, frames = [], [], []

    def record(it, e_local, r):
        energies.append(r.last_energies.copy())
        slots.append(r.slot.copy())
        if (it + 1) % args.reportFrequency == 0:
            frames.append(ctx._ctx.get_state(velocities=False)[0].astype(np.float32))

    start_time = time.time()
    ensemble.run_remd(ctx, remd, exchange_attempts, args.exchange_frequency, callback=record)
    ctx._ctx.synchronize()
    time_taken = time.time() - start_time

    x, v = ctx._ctx.get_state()
    np.savez(my_file, positions=x, velocities=v, slot=remd.slot, iteration=np.int64(remd.iteration), attempted=remd.attempted,
             accepted=remd.accepted, step_count=np.int64(ctx._ctx.step_count), replica_ids=np.arange(remd.local.start, remd.local.stop),
             frames=np.array(frames, dtype=np.float32).reshape(len(frames), R, oa.natoms, 3))
    if rank == 0:
        np.savez(os.path.join(toPath, "output.npz"), temperatures=temps, energies=np.array(energies).reshape(len(energies), remd.n),
                 ladder_index=np.array(slots).reshape(len(slots), remd.n), acceptance=remd.acceptance(),
                 attempted=remd.attempted, accepted=remd.accepted, exchange_frequency=np.int64(args.exchange_frequency),
                 time_step_fs=np.float64(args.timeStep))
        hours, rest = divmod(time_taken, 3600)
        minutes, seconds = divmod(rest, 60)
        print(f"--- {hours} hours {minutes} minutes {seconds} seconds ---")
        print("acceptance per neighbouring pair:", np.array2string(remd.acceptance(), precision=2))
        with open(os.path.join(toPath, "time.dat"), "w") as out:
            out.write(str(time_taken) + "\n")
    return remd


def main(argv=None):
    parser = argparse.ArgumentParser(description="Run OpenAWSEM Replica Exchange MD simulation on the MI355X engine.")
    parser.add_argument("protein", help="The name of the protein")
    parser.add_argument("--to", default="./", help="location of output files")
    parser.add_argument("-c", "--chain", type=str, default="-1")
    parser.add_argument("-t", "--thread", type=int, default=-1, help="accepted for compatibility")
    parser.add_argument("-p", "--platform", type=str, default="HIP")
    parser.add_argument("-r", "--runtime", type=float, default=50, help="total runtime in ns")
    parser.add_argument("--tempStart", type=float, default=200, help="Starting temperature")
    parser.add_argument("--tempEnd", type=float, default=600, help="Ending temperature")
    parser.add_argument("--replicas", type=int, default=12, help="Number of replicas")
    parser.add_argument("-x", "--exchange_frequency", type=int, default=200, help="Time steps between replica exchanges")
    parser.add_argument("--fromCheckPoint", type=str, default=None, help="output folder of the run to continue")
    parser.add_argument("--subMode", type=int, default=-1)
    parser.add_argument("-f", "--forces", default="forces_setup.py")
    parser.add_argument("--parameters", default=None)
    parser.add_argument("--reportFrequency", type=int, default=1, help="iterations between stored frames")
    parser.add_argument("--fromOpenMMPDB", action="store_true", default=False)
    parser.add_argument("--fasta", type=str, default="crystal_structure.fasta")
    parser.add_argument("--timeStep", type=int, default=5, help="Femtosteps per time step, default is 5")
    parser.add_argument("--includeLigands", action="store_true", default=False)
    # extensions of the MI355X engine
    parser.add_argument("--iterations", type=int, default=None, help="number of exchange attempts (overrides --runtime)")
    parser.add_argument("--mixing", choices=["swap-all", "neighbours"], default="swap-all",
                        help="replica mixing scheme per exchange: swap-all = openmmtools' default, which the reference's mm_remd.py runs "
                             "(replicas^3 attempts on uniformly drawn replica pairs); neighbours = one sweep of neighbouring-rung swaps of "
                             "alternating parity (cheaper for long ladders)")
    parser.add_argument("--seed", type=int, default=2026)
    parser.add_argument("--device", type=int, default=None)
    args = parser.parse_args(argv)
    if int(os.environ.get("RANK", 0)) == 0:
        with open("commandline_args.txt", "a") as f:
            f.write(" ".join(sys.argv) + "\n")
    print(" ".join(sys.argv))
    return run_replica_exchange(args)


if __name__ == "__main__":
    main()
