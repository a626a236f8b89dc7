#!/usr/bin/env python3
"""awsem_analyze on the MI355X engine: per-force-group energies of a trajectory -> info.dat, with the command line,
force-group table and output format of openawsem/scripts/mm_analyze.py (:119-125, :139-182).  All frames are
evaluated in batches, one frame per replica slot of a single engine handle."""
import argparse
import os
import sys

import numpy as np

import openawsem_amd
from openawsem_amd import dcd, engine
from openawsem_amd.scripts.mm_run import load_forces_module
from openawsem_amd.unit import femtoseconds, kelvin, picosecond

forceGroupTable = {"Backbone": 20, "Rama": 21, "Contact": 22, "Fragment": 23, "Membrane": 24, "ER": 25, "TBM_Q": 26, "Beta": 27,
                   "Pap": 28, "Helical": 29, "Debye_huckel": 30, "Q": 1, "Rg": 2, "Qc": 3, "Helix_orientation": 18, "Pulling": 19,
                   "Total": list(range(11, 32))}
showValue = ["Q", "Rg"]
showEnergy = ["Backbone", "Rama", "Contact", "Fragment", "Membrane", "ER", "TBM_Q", "Beta", "Pap", "Helical", "Debye_huckel", "Total"]


def read_pdb_trajectory(path):
    frames, cur = [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith(("ATOM", "HETATM")):
                cur.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
            elif line.startswith("ENDMDL") and cur:
                frames.append(cur)
                cur = []
    if cur:
        frames.append(cur)
    return dcd.angstrom_to_nm_f32(np.array(frames, dtype=np.float32)).astype(np.float64)


def analyze(args):
    setupFolderPath = os.path.dirname(args.protein) or "."
    pdb_id = os.path.basename(args.protein)
    chain = args.chain.upper()
    trajectoryPath = os.path.abspath(args.trajectory)
    outFile = os.path.join(os.path.dirname(trajectoryPath), args.output or "info.dat")
    forceSetupFile = os.path.abspath(args.forces) if args.forces else \
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "forces_setup.py")
    parametersLocation = "." if args.parameters is None else os.path.abspath(args.parameters)
    os.chdir(setupFolderPath)
    platform = engine.Platform.getPlatformByName(args.platform)
    if chain == "-1":
        chain = openawsem_amd.getAllChains("crystal_structure.pdb") if os.path.exists("crystal_structure.pdb") else "A"
        print("Chains to simulate: ", chain)
    suffix = "-openmmawsem.pdb"
    input_pdb_filename = pdb_id if (args.fromOpenMMPDB or pdb_id.endswith(suffix)) else f"{pdb_id}{suffix}"
    seq = None if args.fasta == "" else openawsem_amd.read_fasta(args.fasta)

    fileType = trajectoryPath[-3:]
    if fileType == "pdb":
        traj = read_pdb_trajectory(trajectoryPath)
    elif fileType == "dcd":
        traj = dcd.load_dcd_nm(trajectoryPath)
    else:
        raise ValueError(f"Unknown fileType {fileType}")

    oa = openawsem_amd.OpenMMAWSEMSystem(input_pdb_filename, chains=chain, k_awsem=1.0, xml_filename=openawsem_amd.xml,
                                          seqFromPdb=seq, includeLigands=args.includeLigands)
    print(f"using force setup file from {forceSetupFile}")
    forces = load_forces_module(forceSetupFile).set_up_forces(oa, computeQ=not args.noQ, submode=args.subMode,
                                                              contactParameterLocation=parametersLocation)
    oa.addForcesWithDefaultForceGroup(forces)
    batch = max(1, min(len(traj), args.batch))
    integrator = engine.LangevinIntegrator(300 * kelvin, 1 / picosecond, 2 * femtoseconds)
    simulation = engine.Simulation(oa.pdb.topology, oa.system, integrator, platform, n_replicas=batch)
    showAll = showValue + showEnergy
    print("Printing energies")
    with open(outFile, "w") as out:
        line = " ".join(["{0:<8s}".format(i) for i in ["Steps"] + showAll])
        print(line)
        out.write(line + "\n")
        for f0 in range(0, len(traj), batch):
            chunk = traj[f0:f0 + batch]
            for r, x in enumerate(chunk):
                simulation.context.setPositions(x, replica=r)
            eg = simulation.context.energiesByGroup()
            for r in range(len(chunk)):
                e = []
                for term in showAll:
                    g = forceGroupTable[term]
                    val = sum(eg[r, k] for k in g) if isinstance(g, list) else eg[r, g]
                    e.append(val if term in showValue else val / 4.184)
                line = " ".join([f"{f0 + r:<8}"] + ["{0:<8.2f}".format(i) for i in e])
                print(line)
                out.write(line + "\n")
    return outFile


def main(argv=None):
    parser = argparse.ArgumentParser(description="Per-term AWSEM energies of a trajectory on the MI355X engine (awsem_analyze)")
    parser.add_argument("protein", help="The name of the protein")
    parser.add_argument("-d", "--debug", action="store_true", default=False)
    parser.add_argument("-c", "--chain", type=str, default="-1")
    parser.add_argument("--thread", type=int, default=2, help="accepted for compatibility")
    parser.add_argument("-p", "--platform", type=str, default="HIP")
    parser.add_argument("-t", "--trajectory", type=str, default="./movie.pdb")
    parser.add_argument("-o", "--output", type=str, default=None)
    parser.add_argument("--subMode", type=int, default=3)
    parser.add_argument("-f", "--forces", default="forces_setup.py")
    parser.add_argument("--parameters", default=None)
    parser.add_argument("--fromOpenMMPDB", action="store_true", default=False)
    parser.add_argument("--fasta", type=str, default="crystal_structure.fasta")
    parser.add_argument("--includeLigands", action="store_true", default=False)
    parser.add_argument("--batch", type=int, default=256, help="frames evaluated per launch")
    parser.add_argument("--noQ", action="store_true", default=False, help="skip Q / Rg (no crystal_structure-cleaned.pdb needed)")
    args = parser.parse_args(argv)
    with open("analysis_commandline_args.txt", "a") as f:
        f.write(" ".join(sys.argv) + "\n")
    print(" ".join(sys.argv))
    return analyze(args)


if __name__ == "__main__":
    main()
