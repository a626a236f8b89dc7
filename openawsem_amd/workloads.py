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


def build_synthetic(n_chains=122, seed=2026, spacing_nm=5.5, quiet=True, hbonds=False):
    """Synthetic multi-chain system of BASELINE.json configs[4]: `n_chains` copies of the 164-residue 2lzm chain
    (openawsem_amd/data/2lzm: one of the reference's own test structures, tests/data/2lzm-openmmawsem.pdb) on a jittered cubic lattice with
    random rigid rotations (seeded), 122 chains = 20008 residues / 118k atoms.  Terms: con, chain, chi, excl, rama,
    rama_proline, contact (with burial) -- the hydrogen-bond and fragment-memory terms are left out, as the config
    says (contact/burial cell-list scaling); `hbonds=True` adds beta 1-3, P-AP 1-2 and the helical term (dense
    nres x nres tables: meant for a few thousand residues)."""
    import numpy as np
    src = str(data_path.data / "2lzm" / "2lzm-openmmawsem.pdb")
    atoms = [ln for ln in open(src) if ln.startswith("ATOM")]
    xyz = np.array([[float(ln[30:38]), float(ln[38:46]), float(ln[46:54])] for ln in atoms])
    xyz -= xyz.mean(0)
    rng = np.random.default_rng(seed)
    side = int(np.ceil(n_chains ** (1.0 / 3.0)))
    ids = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"
    path = f"/tmp/awsem-synthetic-{n_chains}-{seed}-{os.getpid()}.pdb"
    with open(path, "w") as fh:
        serial = 0
        for c in range(n_chains):
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            w, x, y, z = q
            rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            cell = np.array([c % side, (c // side) % side, c // (side * side)], dtype=float)
            shift = (cell + rng.uniform(-0.1, 0.1, size=3)) * spacing_nm * 10.0
            pos = xyz @ rot.T + shift
            for ln, p in zip(atoms, pos):
                serial += 1
                fh.write("ATOM  %5d %s%s%s%8.3f%8.3f%8.3f%s" % (serial % 100000, ln[12:21], ids[c % len(ids)], ln[22:30], p[0], p[1], p[2], ln[54:]))
            fh.write("TER\n")
        fh.write("END\n")
    seq1 = read_fasta(os.path.join(os.path.dirname(src), "2lzm-crystal_structure.fasta"))
    with contextlib.redirect_stdout(io.StringIO() if quiet else None):
        oa = OpenMMAWSEMSystem(path, k_awsem=1.0, seqFromPdb=seq1 * n_chains)
        oa.addForcesWithDefaultForceGroup([
            basicTerms.con_term(oa), basicTerms.chain_term(oa), basicTerms.chi_term(oa), basicTerms.excl_term(oa, periodic=False),
            basicTerms.rama_term(oa), basicTerms.rama_proline_term(oa), contactTerms.contact_term(oa)])
        if hbonds:
            from .functionTerms import hydrogenBondTerms as hb
            oa.addForcesWithDefaultForceGroup([
                hb.beta_term_1(oa), hb.beta_term_2(oa), hb.beta_term_3(oa), hb.helical_term(oa),
                hb.pap_term_1(oa, ssweightFileName="__no_such_ssweight_file__"), hb.pap_term_2(oa, ssweightFileName="__no_such_ssweight_file__")])
    os.remove(path)
    return oa


def algorithmic_bytes_per_replica_step(oa):
    """fp32 state read+write (x, v: 12 B each way per massive particle = 48 B) plus two
    adjacent fragment-memory table floats per bond (8 B)."""
    n_real = int((oa.system.masses > 0).sum())
    n_fm = sum(f.getNumBonds() for f in oa.system.forces if f.kind == "fragment_memory")
    return n_real * 48 + n_fm * 8, n_real, n_fm
