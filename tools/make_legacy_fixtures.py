#!/usr/bin/env python3
"""Copies the reference's two legacy known-answer tables (and the inputs they need that openawsem_amd/data does not
already hold) into tests/golden/, as DATA: nothing reads /root/reference at run time and no reference source text is
kept.  Run once in the build container; this script is the record of how the fixtures were made.

  tests/golden/legacy_1r69/   from examples/compute_openMM_energy_from_lammps_output/1r69/
      openmmMovie.pdb     30 models of 1r69 (368 atoms each; LAMMPS frames converted by the reference's authors)
      energy_openMM.dat   30 frames x {Con, Chain, Chi, Excluded, Rama, Water, Burial, Fragment, Total}, kcal/mol, %.2f,
                          printed by that folder's computeEnergy.py (OpenCL platform) with frags.mem (1100 memories)
      (the structure, frags.mem + fraglib, gamma.dat and burial_gamma.dat of that folder are byte-identical to / the
       same as what the 1r69-fraglib workload already uses from openawsem_amd/data/1r69 and openawsem_amd/parameters)
  tests/golden/legacy_2xov/   from examples/2xov_complete/
      movie.dcd, info.dat                run0/: 3 frames x 20 columns (kcal/mol; Q, Rg in their own units), %.2f
      membrane_gamma.dat                 setup/2xov_complete/: differs from openawsem/parameters/membrane_gamma.dat
      go_rnativeCACA/CACB/CBCB.dat.gz    setup/2xov_complete/: the er_term's distance tables, gzip of the files as they are
      crystal_structure-ca.pdb           setup/2xov_complete/crystal_structure-cleaned.pdb trimmed to its CA ATOM records
                                         (all that q_value and rg_term read of it)
"""
import gzip
import os
import shutil
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def legacy_1r69():
    src = os.path.join(REF, "examples/compute_openMM_energy_from_lammps_output/1r69")
    dst = os.path.join(OUT, "legacy_1r69")
    os.makedirs(dst, exist_ok=True)
    for name in ("openmmMovie.pdb", "energy_openMM.dat"):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    print("legacy_1r69: copied")


def legacy_2xov():
    ex = os.path.join(REF, "examples/2xov_complete")
    setup = os.path.join(ex, "setup/2xov_complete")
    dst = os.path.join(OUT, "legacy_2xov")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(ex, "run0/movie.dcd"), os.path.join(dst, "movie.dcd"))
    shutil.copy(os.path.join(ex, "run0/info.dat"), os.path.join(dst, "info.dat"))
    shutil.copy(os.path.join(setup, "membrane_gamma.dat"), os.path.join(dst, "membrane_gamma.dat"))
    for name in ("go_rnativeCACA.dat", "go_rnativeCACB.dat", "go_rnativeCBCB.dat"):
        with open(os.path.join(setup, name), "rb") as fin, open(os.path.join(dst, name + ".gz"), "wb") as raw:
            # mtime=0 and no file name in the header: the fixture is reproducible byte for byte
            with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as fout:
                shutil.copyfileobj(fin, fout)
    with open(os.path.join(setup, "crystal_structure-cleaned.pdb")) as fin, open(os.path.join(dst, "crystal_structure-ca.pdb"), "w") as fout:
        for line in fin:
            if line.startswith("ENDMDL"):
                break
            if line.startswith("ATOM") and line[12:16].strip() == "CA":
                fout.write(line)
        fout.write("END\n")
    print("legacy_2xov: copied")


if __name__ == "__main__":
    legacy_1r69()
    legacy_2xov()
