"""The reference's two legacy known-answer tables, as parity pins for BASELINE.json configs[0]/[2] (1r69 with the 1100-memory
fragment library) and configs[3] (the 2xov membrane protein).

Both tables were printed by older driver scripts of the reference's authors (`from openmmawsem import *`) and are kept in its
examples/ folder; tools/make_legacy_fixtures.py copied them (data only) into tests/golden/legacy_*.  They are in kcal/mol with
two decimals, so a column is reproduced when every frame agrees to the printing precision (0.005 + a little fp32 noise: the
1r69 table was computed on OpenMM's single-precision OpenCL platform).

1r69 (examples/compute_openMM_energy_from_lammps_output/1r69/energy_openMM.dat, 30 frames of openmmMovie.pdb):
    Con, Chain, Chi, Excluded, Rama (= rama + rama_proline), Water (= direct + mediated: contact_term without burial),
    Burial (standalone burial_term), Fragment (frags.mem: 1100 memories over 614 fragment files) -- ALL reproduce.
    This pins the many-memory table builder of fragment_memory_term (overlapping fragments, weights, Gly-CB drop), the
    standalone burial_term (NoCutoff) and contact_term(burialPartOn=False), none of which the five golden CSVs cover.

2xov (examples/2xov_complete/run0/info.dat, 3 frames of run0/movie.dcd, forces of run0/forces_setup.py with --subMode 3):
    reproduced: Con, Chain, Chi, Excluded, Rama (rama + rama_proline + rama_ssweight(k = 2 * 8.368)),
                Contact (contact_term(z_dependent=True, inMembrane=True, k_relative_mem=1) incl. burial, with the project
                folder's own membrane_gamma.dat -- the only reference-held numbers for the z-dependent contact variant),
                Helical (z_dependent_helical_term), ER (er_term on go_rnative*.dat), Pap (pap_term_1/2 with the project's
                ssweight; the legacy code's k was 1 kcal/mol where today's default is 0.5 -> k passed explicitly),
                Rg (rg_term, Angstrom) and Q (q_value on the crystal structure);
    reproduced up to a frame-independent constant: Membrane (membrane_preassigned_term(k = 5 kcal/mol, PredictedZim)):
                this engine and the legacy table differ by 32.98 kcal/mol in all three frames (a folded, a relaxed and a
                fully stretched chain, Rg 31 -> 119 A), i.e. by a position-independent term of the legacy code; the
                frame-to-frame differences -- everything a force can see -- are pinned;
    NOT reproduced, and why: Fragment (the legacy run loaded a saved table ./single_frags.npy, a pickle that is not in the
                tree, instead of rebuilding it from single_frags.mem: -1186.18 there, -1380.90 from the .mem file),
                Beta (legacy lambda tables: -36.01 against -109.55 with today's beta_term_1/2/3), Qc / Q_wat / Q_mem /
                Pulling / Rg_Bias (legacy-only terms defined inside the example's forces file), Total (sums the above).
"""
import contextlib
import gzip
import io
import os
import shutil

import numpy as np
import pytest

from openawsem_amd import _capi, data_path, dcd
from openawsem_amd import unit as u
from openawsem_amd.functionTerms import (basicTerms, biasTerms, contactTerms, hydrogenBondTerms, membraneTerms,
                                         templateTerms)
from openawsem_amd.openAWSEM import OpenMMAWSEMSystem, read_fasta
from oracle import awsem_oracle
from tests.common import GOLDEN

KCAL = 4.184
PRINT = 0.005          # half a unit of the last printed digit


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


# ------------------------------------------------------------------------------------------------ 1r69, 30 frames
R69_COLUMNS = ["Con", "Chain", "Chi", "Excluded", "Rama", "Water", "Burial", "Fragment"]
R69_GROUP = {"Con": 11, "Chain": 12, "Chi": 13, "Excluded": 14, "Rama": 15, "Water": 16, "Burial": 17, "Fragment": 19}
# Water is the one column whose worst frame sits outside the rounding of the last digit (0.0058): the table comes from the
# single-precision OpenCL platform and Water sums ~500 pair energies of both signs
R69_TOL = {c: PRINT + 1e-4 for c in R69_COLUMNS}
R69_TOL["Water"] = PRINT + 1.5e-3


def _read_models(path):
    """[(atom name, resSeq)] of the first model and the coordinates of every model, nm (Angstrom * 0.1 in double, as
    openmm.app.PDBFile does)."""
    names, frames, cur = None, [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("ATOM"):
                cur.append((line[12:16].strip(), int(line[22:26]), float(line[30:38]), float(line[38:46]), float(line[46:54])))
            elif line.startswith("END") and cur:
                if names is None:
                    names = [(a[0], a[1]) for a in cur]
                assert [(a[0], a[1]) for a in cur] == names
                frames.append(np.array([a[2:] for a in cur]) * 0.1)
                cur = []
    return names, np.array(frames)


_r69 = {}


def _legacy_1r69(split_burial):
    """(system, frames [30, 368, 3] nm, table [30, 10]).  split_burial: Water and Burial as two forces (contact_term without
    burial + standalone burial_term), as the legacy script had them; otherwise one contact_term in group 16 holding both (the
    cell-list path takes one contact-shaped term)."""
    if split_burial in _r69:
        return _r69[split_burial]
    names, frames = _read_models(os.path.join(GOLDEN, "legacy_1r69", "openmmMovie.pdb"))
    table = np.loadtxt(os.path.join(GOLDEN, "legacy_1r69", "energy_openMM.dat"), skiprows=3)
    assert frames.shape == (30, 368, 3) and table.shape == (30, 10)
    with _cwd(str(data_path.data / "1r69")), contextlib.redirect_stdout(io.StringIO()):
        oa = OpenMMAWSEMSystem("1r69-openmmawsem.pdb", k_awsem=1.0, seqFromPdb=read_fasta("1r69.fasta"))
        assert [(a.name, int(a.residue.id)) for a in oa.pdb.topology.atoms()] == names
        forces = [("Con", basicTerms.con_term(oa)), ("Chain", basicTerms.chain_term(oa)), ("Chi", basicTerms.chi_term(oa)),
                  ("Excluded", basicTerms.excl_term(oa, periodic=False)),
                  ("Rama", basicTerms.rama_term(oa)), ("Rama", basicTerms.rama_proline_term(oa)),
                  ("Water", contactTerms.contact_term(oa, burialPartOn=not split_burial)),
                  ("Fragment", templateTerms.fragment_memory_term(oa, frag_file_list_file="./frags.mem",
                                                                  npy_frag_table=f"/tmp/legacy-frags-{os.getpid()}.npz",
                                                                  UseSavedFragTable=False))]
    if split_burial:
        with _cwd(str(data_path.parameters)), contextlib.redirect_stdout(io.StringIO()):   # burial_term reads ./burial_gamma.dat
            forces.append(("Burial", contactTerms.burial_term(oa)))
    for name, f in forces:
        f.setForceGroup(R69_GROUP[name])
    oa.addForcesWithDefaultForceGroup([f for _, f in forces])
    _r69[split_burial] = (oa, frames, table)
    return _r69[split_burial]


def _check_1r69(e_kj, table, split_burial, extra_tol=0.0):
    e = e_kj / KCAL
    col = {c: table[:, 1 + k] for k, c in enumerate(R69_COLUMNS)}
    for c in ("Con", "Chain", "Chi", "Excluded", "Rama", "Fragment"):
        d = np.abs(e[:, R69_GROUP[c]] - col[c]).max()
        assert d <= R69_TOL[c] + extra_tol, f"{c}: max |diff| = {d:.4f} kcal/mol"
    if split_burial:
        for c in ("Water", "Burial"):
            d = np.abs(e[:, R69_GROUP[c]] - col[c]).max()
            assert d <= R69_TOL[c] + extra_tol, f"{c}: max |diff| = {d:.4f} kcal/mol"
    else:
        d = np.abs(e[:, R69_GROUP["Water"]] - (col["Water"] + col["Burial"])).max()
        assert d <= 2 * PRINT + 1e-4 + extra_tol, f"Water + Burial: max |diff| = {d:.4f} kcal/mol"
    # the table's Total is the sum of its own columns, each rounded on its own
    mine = e[:, sorted(set(R69_GROUP.values()))].sum(1)
    assert np.abs(mine - table[:, 9]).max() <= 8 * PRINT + extra_tol


def test_oracle_reproduces_the_legacy_1r69_table():
    oa, frames, table = _legacy_1r69(True)
    fm = [f for f in oa.system.forces if f.kind == "fragment_memory"][0]
    assert fm.getNumBonds() == 1272            # 'number of bonds: 1272', line 2 of the table's own header
    cpu = awsem_oracle.make_context(oa.system, n_replicas=len(frames), threads=8)
    for r, x in enumerate(frames):
        cpu.set_positions(x, replica=r)
    _check_1r69(cpu.energy_by_group(), table, True)
    # the one-force form the cell-list path is checked with
    oa2, _, _ = _legacy_1r69(False)
    cpu = awsem_oracle.make_context(oa2.system, n_replicas=len(frames), threads=8)
    for r, x in enumerate(frames):
        cpu.set_positions(x, replica=r)
    _check_1r69(cpu.energy_by_group(), table, False)


@pytest.mark.gpu
@pytest.mark.parametrize("cell", [False, True])
@pytest.mark.parametrize("precision", [1, 0])
def test_gpu_reproduces_the_legacy_1r69_table(precision, cell, monkeypatch):
    """fp64 check mode and fp32 production kernels, replica-resident and cell-list path (AWSEM_FORCE_CELL=1)."""
    split = not cell
    oa, frames, table = _legacy_1r69(split)
    if cell:
        monkeypatch.setenv("AWSEM_FORCE_CELL", "1")
    else:   # the split system holds two contact-shaped terms, which only the replica-resident path takes
        monkeypatch.delenv("AWSEM_FORCE_CELL", raising=False)
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=len(frames), precision=precision)
    oa.system.upload(gpu)
    for r, x in enumerate(frames):
        gpu.set_positions(x, replica=r)
    # fp32: 1e-4 relative of energies up to 600 kcal/mol (the north-star tolerance) on top of the printing precision
    _check_1r69(gpu.energy_by_group(), table, split, extra_tol=0.0 if precision == 1 else 2e-3)


# ------------------------------------------------------------------------------------------------ 2xov, 3 frames
XOV_GROUP = {"Q": 1, "Rg": 2, "Con": 11, "Chain": 12, "Chi": 13, "Excluded": 14, "Rama": 15, "Contact": 18, "Membrane": 20,
             "ER": 21, "Pap": 26, "Helical": 28}
XOV_EXACT = ["Con", "Chain", "Chi", "Excluded", "Rama", "Contact", "Helical", "ER", "Pap", "Rg", "Q"]
XOV_MEMBRANE_OFFSET = 32.98      # legacy - this engine, kcal/mol, the same in all three frames (see the module docstring)

_xov = {}


def _legacy_2xov(tmp):
    if "sys" in _xov:
        return _xov["sys"]
    src = os.path.join(GOLDEN, "legacy_2xov")
    data = str(data_path.data / "2xov_complete")
    # a project folder as the legacy run saw it: the package's copy of setup/2xov_complete plus the fixture files
    for name in os.listdir(data):
        shutil.copy(os.path.join(data, name), os.path.join(tmp, name))
    for name in ("go_rnativeCACA.dat", "go_rnativeCACB.dat", "go_rnativeCBCB.dat"):
        with gzip.open(os.path.join(src, name + ".gz"), "rb") as fin, open(os.path.join(tmp, name), "wb") as fout:
            shutil.copyfileobj(fin, fout)
    shutil.copy(os.path.join(src, "membrane_gamma.dat"), os.path.join(tmp, "membrane_gamma.dat"))
    for name in ("gamma.dat", "burial_gamma.dat"):       # byte-identical in the project folder and in the package
        shutil.copy(str(data_path.parameters / name), os.path.join(tmp, name))
    shutil.copy(os.path.join(src, "crystal_structure-ca.pdb"), os.path.join(tmp, "crystal_structure-cleaned.pdb"))
    kcal = u.kilocalorie_per_mole
    with _cwd(tmp), contextlib.redirect_stdout(io.StringIO()):
        oa = OpenMMAWSEMSystem("2xov_complete-openmmawsem.pdb", k_awsem=1.0, seqFromPdb=read_fasta("2xov_complete.fasta"))
        forces = [
            ("Con", basicTerms.con_term(oa)), ("Chain", basicTerms.chain_term(oa)), ("Chi", basicTerms.chi_term(oa)),
            ("Excluded", basicTerms.excl_term(oa, periodic=False)),
            ("Rama", basicTerms.rama_term(oa)), ("Rama", basicTerms.rama_proline_term(oa)),
            ("Rama", basicTerms.rama_ssweight_term(oa, k_rama_ssweight=2 * 8.368)),
            ("Pap", hydrogenBondTerms.pap_term_1(oa, k=1 * kcal)), ("Pap", hydrogenBondTerms.pap_term_2(oa, k=1 * kcal)),
            ("Membrane", membraneTerms.membrane_preassigned_term(oa, k=5 * kcal, membrane_center=0 * u.angstrom, zimFile="PredictedZim")),
            ("Helical", hydrogenBondTerms.z_dependent_helical_term(oa, k_helical=1 * 4.184, membrane_center=0 * u.angstrom)),
            ("ER", templateTerms.er_term(oa)),
            ("Contact", contactTerms.contact_term(oa, k_contact=1 * 4.184, z_dependent=True, inMembrane=True,
                                                  membrane_center=0 * u.angstrom, k_relative_mem=1, periodic=False,
                                                  parametersLocation=".", burialPartOn=True)),
            ("Rg", biasTerms.rg_term(oa)), ("Q", biasTerms.q_value(oa, "crystal_structure-cleaned.pdb")),
        ]
    for name, f in forces:
        f.setForceGroup(XOV_GROUP[name])
    oa.addForcesWithDefaultForceGroup([f for _, f in forces])
    frames = dcd.load_dcd_nm(os.path.join(src, "movie.dcd"))
    with open(os.path.join(src, "info.dat")) as fh:
        header = fh.readline().split()
    table = np.loadtxt(os.path.join(src, "info.dat"), skiprows=1)
    assert frames.shape == (3, 1620, 3) and table.shape == (3, 21)
    _xov["sys"] = (oa, frames, {c: table[:, k] for k, c in enumerate(header)})
    return _xov["sys"]


def _check_2xov(e_kj, col, extra_rel=0.0):
    for c in XOV_EXACT:
        mine = e_kj[:, XOV_GROUP[c]] / (1.0 if c in ("Q", "Rg") else KCAL)
        tol = PRINT + 1e-4 + extra_rel * np.abs(col[c])
        d = np.abs(mine - col[c])
        assert (d <= tol).all(), f"{c}: |diff| = {d} (kcal/mol; Q, Rg in their own units)"
    m = e_kj[:, XOV_GROUP["Membrane"]] / KCAL
    assert np.abs(col["Membrane"] - m + XOV_MEMBRANE_OFFSET).max() <= 2 * PRINT + 1e-4 + extra_rel * 600
    dm, dl = m - m[0], col["Membrane"] - col["Membrane"][0]
    assert np.abs(dm - dl).max() <= 2 * PRINT + 1e-4 + extra_rel * 600


@pytest.fixture(scope="module")
def xov_folder(tmp_path_factory):
    return str(tmp_path_factory.mktemp("legacy_2xov"))


def test_oracle_reproduces_the_legacy_2xov_table(xov_folder):
    oa, frames, col = _legacy_2xov(xov_folder)
    cpu = awsem_oracle.make_context(oa.system, n_replicas=3, threads=3)
    for r, x in enumerate(frames):
        cpu.set_positions(x, replica=r)
    _check_2xov(cpu.energy_by_group(), col)


@pytest.mark.gpu
@pytest.mark.parametrize("cell", [False, True])
@pytest.mark.parametrize("precision", [1, 0])
def test_gpu_reproduces_the_legacy_2xov_table(precision, cell, xov_folder, monkeypatch):
    oa, frames, col = _legacy_2xov(xov_folder)
    if cell:
        monkeypatch.setenv("AWSEM_FORCE_CELL", "1")
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=3, precision=precision)
    oa.system.upload(gpu)
    for r, x in enumerate(frames):
        gpu.set_positions(x, replica=r)
    # fp32: the north-star tolerance, 1e-4 relative, on top of the printing precision (frame 0 is the unrelaxed crystal
    # structure: Con, Chain, Chi are 1700-2000 kcal/mol there)
    _check_2xov(gpu.energy_by_group(), col, extra_rel=0.0 if precision == 1 else 1e-4)
