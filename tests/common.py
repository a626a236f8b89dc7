"""Shared helpers: build the reference test's force list (tests/test_energies.py:56-75 of the reference) for a
golden protein and load its trajectory the way the reference harness does."""
from __future__ import annotations

import contextlib
import csv
import functools
import io
import os

import numpy as np

import openawsem_amd as oa_pkg
from openawsem_amd import dcd
from openawsem_amd.functionTerms import (basicTerms, contactTerms, debyeHuckelTerms, hydrogenBondTerms,
                                         templateTerms)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROTEINS = ["1brs", "1mbn", "1ubq", "2lyz", "2lzm"]
COLUMNS = ["Backbone", "Rama", "Contact", "Fragment", "Membrane", "ER", "TBM_Q", "Beta", "Pap", "Helical"]
FORCE_GROUP = {"Backbone": 20, "Rama": 21, "Contact": 22, "Fragment": 23, "Membrane": 24, "ER": 25, "TBM_Q": 26,
               "Beta": 27, "Pap": 28, "Helical": 29, "DebyeHuckel": 30}


def golden_forces(oa, protein, tmpdir=None):
    """The 16 forces of the reference test, keyed by the reference's benchmark names."""
    g = lambda name: os.path.join(GOLDEN, name)  # noqa: E731
    npy = os.path.join(tmpdir or "/tmp", f"{protein}-single_frags-{os.getpid()}.npz")
    with contextlib.redirect_stdout(io.StringIO()):
        forces = {
            "Backbone": basicTerms.con_term(oa),
            "Rama": basicTerms.rama_term(oa),
            "Contact": contactTerms.contact_term(oa),
            "Chain": basicTerms.chain_term(oa),
            "Chi": basicTerms.chi_term(oa),
            "Excluded": basicTerms.excl_term(oa, periodic=False),
            "RamaProline": basicTerms.rama_proline_term(oa),
            "RamaSSWeight": basicTerms.rama_ssweight_term(oa, k_rama_ssweight=2 * 8.368, ssweight_file=g(f"{protein}-ssweight")),
            "Beta1": hydrogenBondTerms.beta_term_1(oa),
            "Beta2": hydrogenBondTerms.beta_term_2(oa),
            "Beta3": hydrogenBondTerms.beta_term_3(oa),
            "Helical": hydrogenBondTerms.helical_term(oa),
            "Pap1": hydrogenBondTerms.pap_term_1(oa, ssweightFileName="__no_such_ssweight_file__"),
            "Pap2": hydrogenBondTerms.pap_term_2(oa, ssweightFileName="__no_such_ssweight_file__"),
            "FragmentMemory": templateTerms.fragment_memory_term(
                oa, frag_file_list_file=g(f"{protein}-single_frags.mem"), npy_frag_table=npy, UseSavedFragTable=False),
            "DebyeHuckel": debyeHuckelTerms.debye_huckel_term(oa, chargeFile=g(f"{protein}-charge.txt")),
        }
    if os.path.exists(npy):
        os.remove(npy)
    return forces


@functools.lru_cache(maxsize=None)
def golden_system(protein):
    """(OpenMMAWSEMSystem with all 16 test forces added, trajectory [12, natoms, 3] nm, golden table dict)."""
    seq = oa_pkg.read_fasta(os.path.join(GOLDEN, f"{protein}-crystal_structure.fasta"))
    oa = oa_pkg.OpenMMAWSEMSystem(os.path.join(GOLDEN, f"{protein}-openmmawsem.pdb"), k_awsem=1.0, seqFromPdb=seq)
    oa.addForcesWithDefaultForceGroup(list(golden_forces(oa, protein).values()))
    traj = dcd.load_dcd_nm(os.path.join(GOLDEN, f"{protein}-movie.dcd"))
    with open(os.path.join(GOLDEN, f"{protein}_energies.csv")) as fh:
        rows = list(csv.DictReader(fh))
    table = {c: np.array([float(r[c]) for r in rows]) for c in COLUMNS}
    return oa, traj, table


def energies_by_column(ctx, traj):
    """Per-frame energies of the reference's COLUMNS from one context (frames set one at a time on replica 0..)."""
    out = {c: [] for c in COLUMNS}
    R = ctx.n_replicas
    for f0 in range(0, len(traj), R):
        chunk = traj[f0:f0 + R]
        for r, x in enumerate(chunk):
            ctx.set_positions(x, replica=r)
        e = ctx.energy_by_group()
        for c in COLUMNS:
            out[c].extend(e[:len(chunk), FORCE_GROUP[c]])
    return {c: np.array(v) for c, v in out.items()}
