#!/usr/bin/env python3
"""awsem_run on the MI355X engine: same command line, files and schedule as openawsem/scripts/mm_run.py
(argparse flags :219-246; annealing loop :169-175; outputs native.pdb, movie.dcd, output.log, checkpnt.chk, time.dat),
plus --replicas / --seed for a batched ensemble of independent trajectories in one engine handle."""
import argparse
import importlib.util
import os
import shutil
import sys
import time

import numpy as np

import openawsem_amd
from openawsem_amd import engine
from openawsem_amd.unit import femtoseconds, kelvin, picosecond


def load_forces_module(path):
    openawsem_amd.install_as_openawsem()
    spec = importlib.util.spec_from_file_location("forces", path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def run(args):
    platform = engine.Platform.getPlatformByName(args.platform)
    if args.platform == "CPU" and args.thread != -1:
        platform.setPropertyDefaultValue("Threads", str(args.thread))
    if args.device is not None:
        platform.setPropertyDefaultValue("DeviceIndex", str(args.device))

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
    if args.to != "./":
        os.makedirs(toPath, exist_ok=True)
        shutil.copy(forceSetupFile, os.path.join(toPath, "forces_setup.py"))
        for f in ("crystal_structure.fasta", "crystal_structure.pdb"):
            if os.path.exists(f):
                shutil.copy(f, toPath)

    suffix = "-openmmawsem.pdb"
    if args.fromOpenMMPDB:
        input_pdb_filename = pdb_id
    else:
        input_pdb_filename = pdb_id if pdb_id.endswith(suffix) else f"{pdb_id}{suffix}"
    seq = None if args.fasta == "" else openawsem_amd.read_fasta(args.fasta)
    if seq is not None:
        print(f"Using Seq:\n{seq}")

    snapShotCount = 400
    stepsPerT = int(args.steps / snapShotCount)
    Tstart, Tend = args.tempStart, args.tempEnd
    reporter_frequency = args.reportFrequency if args.reportFrequency != -1 else (stepsPerT if stepsPerT else 4000)

    print(f"using force setup file from {forceSetupFile}")
    forces = load_forces_module(forceSetupFile)
    oa = openawsem_amd.OpenMMAWSEMSystem(input_pdb_filename, k_awsem=1.0, chains=chain, xml_filename=openawsem_amd.xml,
                                          seqFromPdb=seq, includeLigands=args.includeLigands)
    myForces = forces.set_up_forces(oa, submode=args.subMode, contactParameterLocation=parametersLocation)
    oa.addForcesWithDefaultForceGroup(myForces)

    R = args.replicas
    integrator = engine.LangevinIntegrator(Tstart * kelvin, 1 / picosecond, args.timeStep * femtoseconds)
    integrator.setRandomNumberSeed(args.seed)
    simulation = engine.Simulation(oa.pdb.topology, oa.system, integrator, platform, n_replicas=R)
    if args.fromCheckPoint:
        simulation.loadCheckpoint(checkPointPath)
    else:
        simulation.context.setPositions(oa.pdb.positions)
        # the reference writes the input and the minimised structure as the first two frames (mm_run.py:130-137)
        native = engine.PDBReporter(os.path.join(toPath, "native.pdb"), 1)
        movie0 = engine.DCDReporter(os.path.join(toPath, "movie.dcd"), 1)
        native.report(simulation), movie0.report(simulation)
        if not args.noMinimize:
            simulation.minimizeEnergy()                   # all replicas on the device (identical start -> identical minimum)
        native.report(simulation), movie0.report(simulation)
        simulation.context.setVelocitiesToTemperature(Tstart * kelvin, args.seed)

    print("reporter_frequency", reporter_frequency)
    simulation.reporters.append(engine.StateDataReporter(sys.stdout, reporter_frequency, step=True, potentialEnergy=True, temperature=True))
    simulation.reporters.append(engine.StateDataReporter(os.path.join(toPath, "output.log"), reporter_frequency, step=True,
                                                         potentialEnergy=True, temperature=True))
    simulation.reporters.append(engine.DCDReporter(os.path.join(toPath, "movie.dcd"), reporter_frequency, append=True))
    simulation.reporters.append(engine.CheckpointReporter(os.path.join(toPath, "checkpnt.chk"), 1000))

    print("Simulation Starts")
    start_time = time.time()
    if args.simulation_mode == 0:
        simulation.step(int(args.steps))
    elif args.simulation_mode == 1:
        deltaT = (Tend - Tstart) / snapShotCount
        for i in range(snapShotCount):
            integrator.setTemperature((Tstart + deltaT * i) * kelvin)
            simulation.step(stepsPerT)
    simulation.flushReporters()            # the reporters' files are complete (they are written behind the launches: engine.Simulation.step)
    simulation.context._ctx.synchronize()
    time_taken = time.time() - start_time
    hours, rest = divmod(time_taken, 3600)
    minutes, seconds = divmod(rest, 60)
    print(f"---{hours} hours {minutes} minutes {seconds} seconds ---")
    with open(os.path.join(toPath, "time.dat"), "w") as out:
        out.write(str(time_taken) + "\n")
    if R > 1:
        c = simulation.context
        np.savez(os.path.join(toPath, "ensemble_final.npz"), energies=c.energies(),
                 positions=c._ctx.get_state(velocities=False)[0])
    return simulation


def main(argv=None):
    parser = argparse.ArgumentParser(description="Run an AWSEM simulation on the MI355X engine (awsem_run)")
    parser.add_argument("protein", help="The name of the protein")
    parser.add_argument("--name", default="simulation", help="Name of the simulation")
    parser.add_argument("--to", default="./", help="location of movie file")
    parser.add_argument("-c", "--chain", type=str, default="-1")
    parser.add_argument("-t", "--thread", type=int, default=-1, help="accepted for compatibility")
    parser.add_argument("-p", "--platform", type=str, default="HIP", help="HIP (fp32) or Reference (fp64 check mode); OpenCL/CUDA/CPU map to HIP")
    parser.add_argument("-s", "--steps", type=float, default=2e4, help="number of steps")
    parser.add_argument("--tempStart", type=float, default=800, help="Starting temperature")
    parser.add_argument("--tempEnd", type=float, default=200, help="Ending temperature")
    parser.add_argument("--fromCheckPoint", type=str, default=None, help="The checkpoint file you want to start from")
    parser.add_argument("-m", "--simulation_mode", type=int, default=1, help="0: constant temperature, 1: temperature annealing")
    parser.add_argument("--subMode", type=int, default=-1)
    parser.add_argument("-f", "--forces", default="forces_setup.py")
    parser.add_argument("--parameters", default=None)
    parser.add_argument("-r", "--reportFrequency", type=int, default=-1, help="default value step/400")
    parser.add_argument("--fromOpenMMPDB", action="store_true", default=False)
    parser.add_argument("--fasta", type=str, default="crystal_structure.fasta")
    parser.add_argument("--timeStep", type=int, default=2)
    parser.add_argument("--includeLigands", action="store_true", default=False)
    # extensions of the MI355X engine
    parser.add_argument("--replicas", type=int, default=1, help="independent trajectories batched on the GPU")
    parser.add_argument("--seed", type=int, default=2026)
    parser.add_argument("--device", type=int, default=None)
    parser.add_argument("--noMinimize", action="store_true", default=False)
    args = parser.parse_args(argv)
    with open("commandline_args.txt", "a") as f:
        f.write(" ".join(sys.argv) + "\n")
    print(" ".join(sys.argv))
    return run(args)


if __name__ == "__main__":
    main()
