"""Named workloads of BASELINE.json built from the inputs shipped under openawsem_amd/data (copied from the
reference's examples/ so that nothing reads /root/reference at run time)."""
from __future__ import annotations

import contextlib
import io
import os

from . import data_path
from .functionTerms import (basicTerms, contactTerms, debyeHuckelTerms, hydrogenBondTerms, membraneTerms,
                            templateTerms)
from .openAWSEM import OpenMMAWSEMSystem, read_fasta


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


def default_forces(oa):
    """The default term list of the reference's openawsem/scripts/forces_setup.py:33-66, evaluated in the project
    folder (cwd) like awsem_run does: con, chain, chi, excl, rama, rama_proline, contact, beta1-3, pap1-2,
    fragment memory (single_frags.mem), Debye-Hueckel (charge.txt when present, else charges from the sequence)."""
    charge = "charge.txt" if os.path.exists("charge.txt") else None
    return [
        basicTerms.con_term(oa),
        basicTerms.chain_term(oa),
        basicTerms.chi_term(oa),
        basicTerms.excl_term(oa, periodic=False),
        basicTerms.rama_term(oa),
        basicTerms.rama_proline_term(oa),
        contactTerms.contact_term(oa),
        hydrogenBondTerms.beta_term_1(oa),
        hydrogenBondTerms.beta_term_2(oa),
        hydrogenBondTerms.beta_term_3(oa),
        hydrogenBondTerms.pap_term_1(oa),
        hydrogenBondTerms.pap_term_2(oa),
        templateTerms.fragment_memory_term(oa, frag_file_list_file="./single_frags.mem",
                                           npy_frag_table=f"/tmp/single_frags-{os.getpid()}.npz", UseSavedFragTable=False),
        debyeHuckelTerms.debye_huckel_term(oa, chargeFile=charge),
    ]


def _fragment_memory(oa, mem_file):
    tmp = f"/tmp/frags-{os.getpid()}.npz"
    fm = templateTerms.fragment_memory_term(oa, frag_file_list_file=mem_file, npy_frag_table=tmp, UseSavedFragTable=False)
    if os.path.exists(tmp):
        os.remove(tmp)
    return fm


def build_1r69(quiet=True, fragment_library=False):
    """1r69 (63 residues, 368 atoms) from its extended-chain openmmawsem.pdb with the default forces_setup.py terms:
    BASELINE.json configs[0]/[1].  fragment_library=True swaps the single native memory for the reference's
    examples/1r69/frags.mem (1100 memories over 614 fragment files, trimmed to the atoms the term reads by
    tools/make_workload_fixtures.py): the many-memory stress named in BASELINE.json configs[2]."""
    folder = str(data_path.data / "1r69")
    with _cwd(folder), contextlib.redirect_stdout(io.StringIO() if quiet else None):
        oa = OpenMMAWSEMSystem("1r69-openmmawsem.pdb", k_awsem=1.0, seqFromPdb=read_fasta("1r69.fasta"))
        forces = default_forces(oa)
        if fragment_library:
            forces[12] = _fragment_memory(oa, "./frags.mem")
        oa.addForcesWithDefaultForceGroup(forces)
    tmp = f"/tmp/single_frags-{os.getpid()}.npz"
    if os.path.exists(tmp):
        os.remove(tmp)
    return oa


def build_t0958(quiet=True):
    """T0958 (96 residues, 559 atoms): default terms + helical_term, fragment memory from single_frags.mem (the
    reference's examples/T0958 ships no fragment library, so the memory is its own native structure, see
    tools/make_workload_fixtures.py); pap terms read the project's ssweight: BASELINE.json configs[2]."""
    folder = str(data_path.data / "T0958")
    with _cwd(folder), contextlib.redirect_stdout(io.StringIO() if quiet else None):
        oa = OpenMMAWSEMSystem("t0958-openmmawsem.pdb", k_awsem=1.0, seqFromPdb=read_fasta("t0958.fasta"))
        forces = default_forces(oa)
        forces.append(hydrogenBondTerms.helical_term(oa))
        forces.append(basicTerms.rama_ssweight_term(oa, k_rama_ssweight=2 * 8.368, ssweight_file="ssweight"))
        oa.addForcesWithDefaultForceGroup(forces)
    tmp = f"/tmp/single_frags-{os.getpid()}.npz"
    if os.path.exists(tmp):
        os.remove(tmp)
    return oa


def build_2xov(quiet=True):
    """2xov_complete (276 residues) membrane protein: default terms with the membrane variant of the contact term
    plus membrane_term (reference forces_setup.py:46,56): BASELINE.json configs[3]."""
    from . import unit as u
    folder = str(data_path.data / "2xov_complete")
    with _cwd(folder), contextlib.redirect_stdout(io.StringIO() if quiet else None):
        oa = OpenMMAWSEMSystem("2xov_complete-openmmawsem.pdb", k_awsem=1.0, seqFromPdb=read_fasta("2xov_complete.fasta"))
        forces = default_forces(oa)
        forces[6] = contactTerms.contact_term(oa, z_dependent=True, inMembrane=True, membrane_center=0 * u.angstrom,
                                              k_relative_mem=3)
        forces.append(membraneTerms.membrane_term(oa, k=1 * u.kilocalorie_per_mole, membrane_center=0 * u.angstrom))
        oa.addForcesWithDefaultForceGroup(forces)
    tmp = f"/tmp/single_frags-{os.getpid()}.npz"
    if os.path.exists(tmp):
        os.remove(tmp)
    return oa


def algorithmic_bytes_per_replica_step(oa):
    """fp32 state read+write (x, v: 12 B each way per massive particle = 48 B) plus two
    adjacent fragment-memory table floats per bond (8 B)."""
    n_real = int((oa.system.masses > 0).sum())
    n_fm = sum(f.getNumBonds() for f in oa.system.forces if f.kind == "fragment_memory")
    return n_real * 48 + n_fm * 8, n_real, n_fm
