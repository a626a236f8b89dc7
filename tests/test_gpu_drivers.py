"""awsem_run / awsem_analyze on the MI355X engine, driven exactly like the reference's command-line tools
(openawsem/scripts/mm_run.py, mm_analyze.py): a project folder, a user forces_setup.py written against the
reference's module names (`from openawsem.functionTerms import *`), files out.

awsem_analyze is pinned by the reference's golden energies (tests/data/<p>_energies.csv, kJ/mol -> info.dat prints
kcal/mol with 2 decimals).  awsem_run has no reference fixture (trajectories depend on OpenMM's RNG): its outputs are
checked for presence, shape and physical sanity ("parity unpinned").
"""
import os
import shutil

import numpy as np
import pytest

from openawsem_amd import data_path, dcd
from tests.common import COLUMNS, GOLDEN, golden_system

pytestmark = pytest.mark.gpu

# the reference test's force list (tests/test_energies.py:56-75) as a user forces file in the reference's own idiom
FORCES_SETUP = '''
from openawsem.functionTerms import *
from openawsem.helperFunctions.myFunctions import *

try:
    from openmm import Force
    from openmm.unit import angstrom, Quantity
except ModuleNotFoundError:
    from simtk.openmm import Force
    from simtk.unit import angstrom, Quantity
from typing import List
from openawsem.openAWSEM import OpenMMAWSEMSystem


def set_up_forces(oa: OpenMMAWSEMSystem, computeQ: bool = False, submode: int = -1, contactParameterLocation: str = ".",
                  membrane_center: Quantity = -0*angstrom) -> List[Force]:
    forces = [
        basicTerms.con_term(oa),
        basicTerms.chain_term(oa),
        basicTerms.chi_term(oa),
        basicTerms.excl_term(oa, periodic=False),
        basicTerms.rama_term(oa),
        basicTerms.rama_proline_term(oa),
        basicTerms.rama_ssweight_term(oa, k_rama_ssweight=2*8.368, ssweight_file="ssweight_for_rama"),
        contactTerms.contact_term(oa),
        hydrogenBondTerms.beta_term_1(oa),
        hydrogenBondTerms.beta_term_2(oa),
        hydrogenBondTerms.beta_term_3(oa),
        hydrogenBondTerms.helical_term(oa),
        hydrogenBondTerms.pap_term_1(oa),
        hydrogenBondTerms.pap_term_2(oa),
        templateTerms.fragment_memory_term(oa, frag_file_list_file="./single_frags.mem", npy_frag_table="./single_frags.npy", UseSavedFragTable=False),
        debyeHuckelTerms.debye_huckel_term(oa, chargeFile="charge.txt"),
    ]
    if computeQ:      # scripts/forces_setup.py:67-69 of the reference
        forces.append(biasTerms.rg_term(oa))
        forces.append(biasTerms.q_value(oa, "crystal_structure-cleaned.pdb", forceGroup=1))
    return forces
'''


def _golden_project(tmp_path, protein):
    d = tmp_path / protein
    d.mkdir()
    shutil.copy(os.path.join(GOLDEN, f"{protein}-openmmawsem.pdb"), d / f"{protein}-openmmawsem.pdb")
    shutil.copy(os.path.join(GOLDEN, f"{protein}-crystal_structure.fasta"), d / "crystal_structure.fasta")
    shutil.copy(os.path.join(GOLDEN, f"{protein}-crystal_structure.pdb"), d / "crystal_structure.pdb")
    shutil.copy(os.path.join(GOLDEN, f"{protein}-movie.dcd"), d / "movie.dcd")
    shutil.copy(os.path.join(GOLDEN, f"{protein}-single_frags.mem"), d / "single_frags.mem")
    shutil.copy(os.path.join(GOLDEN, f"{protein}-charge.txt"), d / "charge.txt")
    shutil.copy(os.path.join(GOLDEN, f"{protein}-ssweight"), d / "ssweight_for_rama")    # no file named "ssweight": pap terms see zeros, as in the reference test
    for f in os.listdir(GOLDEN):
        if f.startswith(protein + "_") and f.endswith(".gro"):
            shutil.copy(os.path.join(GOLDEN, f), d / f)
    (d / "forces_setup.py").write_text(FORCES_SETUP)
    return d


@pytest.mark.parametrize("protein", ["1ubq", "1brs"])
def test_awsem_analyze_reproduces_golden_energies(protein, tmp_path, monkeypatch):
    from openawsem_amd.scripts import mm_analyze
    d = _golden_project(tmp_path, protein)
    monkeypatch.chdir(d)
    out = mm_analyze.main([protein, "-t", "movie.dcd", "-f", "forces_setup.py", "--noQ", "-p", "Reference"])
    rows = [ln.split() for ln in open(out).read().strip().split("\n")]
    header, data = rows[0], np.array(rows[1:], dtype=float)
    assert header[:3] == ["Steps", "Q", "Rg"] and len(data) == 12
    _, _, golden = golden_system(protein)
    for c in COLUMNS:
        col = header.index(c)
        assert np.abs(data[:, col] - golden[c] / 4.184).max() <= 0.0051, c      # 2 printed decimals, kcal/mol
    total = sum(golden[c] for c in COLUMNS)
    dh = data[:, header.index("Debye_huckel")]
    assert np.abs(data[:, header.index("Total")] - (total / 4.184 + dh)).max() <= 0.05


def test_awsem_analyze_prints_q_and_rg_from_the_device(tmp_path, monkeypatch):
    """Q and Rg are force-group "energies" (groups 1 and 2) as in the reference; no fixture pins them (parity unpinned):
    compared with a direct numpy evaluation of biasTerms.py:13-83, 146-188 and 461-486."""
    from openawsem_amd.scripts import mm_analyze
    d = _golden_project(tmp_path, "1ubq")
    shutil.copy(d / "crystal_structure.pdb", d / "crystal_structure-cleaned.pdb")
    monkeypatch.chdir(d)
    out = mm_analyze.main(["1ubq", "-t", "movie.dcd", "-f", "forces_setup.py", "-p", "Reference"])
    rows = [ln.split() for ln in open(out).read().strip().split("\n")]
    header, data = rows[0], np.array(rows[1:], dtype=float)
    oa, traj, golden = golden_system("1ubq")
    nat = []
    for line in open(d / "crystal_structure.pdb"):
        if line.startswith("ATOM") and line[12:16].strip() == "CA":
            nat.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    nat = np.array(nat) / 10.0
    i, j = np.nonzero(np.abs(np.subtract.outer(np.arange(oa.nres), np.arange(oa.nres))) >= 3)
    r0, sig = np.linalg.norm(nat[i] - nat[j], axis=1), 0.1 * np.abs(i - j) ** 0.15
    for f in range(12):
        ca = traj[f][oa.ca]
        q = np.exp(-(np.linalg.norm(ca[i] - ca[j], axis=1) - r0) ** 2 / (2 * sig ** 2)).mean()
        rg = 10 * np.sqrt(((ca[:, None] - ca[None]) ** 2).sum(-1).sum() / 2) / oa.nres
        assert abs(data[f, header.index("Q")] - q) <= 0.0051 and abs(data[f, header.index("Rg")] - rg) <= 0.0051
    assert data[0, header.index("Q")] > 0.9 > data[11, header.index("Q")]
    assert np.abs(data[:, header.index("Contact")] - golden["Contact"] / 4.184).max() <= 0.0051     # the other columns are untouched


def test_awsem_run_anneals_an_ensemble_and_writes_the_reference_files(tmp_path, monkeypatch):
    from openawsem_amd.scripts import mm_run
    src = data_path.data / "1r69"
    d = tmp_path / "1r69"
    shutil.copytree(src, d, ignore=shutil.ignore_patterns("fraglib", "frags.mem"))
    shutil.copy(d / "1r69.fasta", d / "crystal_structure.fasta")
    seq = "".join(ln.strip() for ln in open(d / "1r69.fasta") if not ln.startswith(">"))
    # charge.txt as awsem_create writes it (mm_create_project.py: one "index charge" line per residue)
    (d / "charge.txt").write_text("".join(f"{i} {1.0 if a in 'RK' else (-1.0 if a in 'DE' else 0.0)}\n" for i, a in enumerate(seq)))
    monkeypatch.chdir(d)
    out = tmp_path / "run0"
    sim = mm_run.main(["1r69", "--to", str(out), "-s", "2000", "--tempStart", "600", "--tempEnd", "300", "--replicas", "8", "--seed", "7",
                       "-f", os.path.join(os.path.dirname(mm_run.__file__), "forces_setup.py")])
    for f in ("native.pdb", "movie.dcd", "output.log", "checkpnt.chk", "time.dat", "forces_setup.py", "ensemble_final.npz"):
        assert (out / f).exists(), f
    movie = dcd.load_dcd_nm(str(out / "movie.dcd"))
    assert movie.shape == (2 + 400, 368, 3) and np.isfinite(movie).all()      # input + minimised + one frame per stage
    log = np.loadtxt(out / "output.log", delimiter=",", skiprows=1)
    assert log.shape == (400, 3) and log[-1, 0] == 2000
    assert 150 < log[-20:, 2].mean() < 500                                     # kinetic temperature follows the schedule
    assert log[0, 1] < 0 or log[:, 1].min() < log[0, 1]                        # minimised / annealed below the start energy
    with np.load(out / "ensemble_final.npz") as z:
        assert z["energies"].shape == (8,) and np.isfinite(z["energies"]).all()
        assert np.abs(z["positions"][0] - z["positions"][1]).max() > 1e-3      # replicas are independent trajectories
    # restart from the checkpoint the run left
    monkeypatch.chdir(d)
    sim2 = mm_run.main(["1r69", "--to", str(tmp_path / "run1"), "-s", "400", "-m", "0", "--tempStart", "300", "--replicas", "8",
                        "--fromCheckPoint", str(out / "checkpnt.chk"), "-f", os.path.join(os.path.dirname(mm_run.__file__), "forces_setup.py")])
    assert sim2.currentStep == 2000 + 400


@pytest.mark.parametrize("cell", [False, True])
def test_reports_written_behind_the_launches_are_byte_identical(cell, tmp_path, monkeypatch):
    """engine.Simulation.step serves the built-in reporters in two halves -- values fetched asynchronously behind the launch
    (awsem_snapshot_begin, pinned host memory), files written by a worker thread after the next launches have been queued -- where the
    reference's OpenMM reporters run synchronously inside simulation.step (scripts/mm_run.py:157-175).  Same values, same order: every
    file of a run must equal, byte for byte, what AWSEM_ASYNC_REPORT=0 (synchronous reporting) writes.
    cell: the same on the cell-list path, whose steps are captured HIP graphs -- the worker thread's wait on a snapshot event must not
    fall into a capture on the same stream (it invalidated the capture until the two were serialised: Engine::capture_mu_)."""
    from openawsem_amd.scripts import mm_run
    if cell:
        monkeypatch.setenv("AWSEM_FORCE_CELL", "1")
    d = tmp_path / "1r69"
    shutil.copytree(data_path.data / "1r69", d, ignore=shutil.ignore_patterns("fraglib", "frags.mem"))
    shutil.copy(d / "1r69.fasta", d / "crystal_structure.fasta")
    seq = "".join(ln.strip() for ln in open(d / "1r69.fasta") if not ln.startswith(">"))
    (d / "charge.txt").write_text("".join(f"{i} {1.0 if a in 'RK' else (-1.0 if a in 'DE' else 0.0)}\n" for i, a in enumerate(seq)))
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("AWSEM_ASYNC_REPORT", mode)
        monkeypatch.chdir(d)
        out = tmp_path / f"run_async{mode}"
        sim = mm_run.main(["1r69", "--to", str(out), "-s", "4000", "--tempStart", "700", "--tempEnd", "300", "--replicas", "16", "--seed", "11",
                           "-f", os.path.join(os.path.dirname(mm_run.__file__), "forces_setup.py")])
        assert (sim._worker is not None) == (mode == "1")
        outs.append(out)
    monkeypatch.delenv("AWSEM_ASYNC_REPORT", raising=False)
    for f in ("native.pdb", "movie.dcd", "output.log", "checkpnt.chk"):
        a, b = (outs[0] / f).read_bytes(), (outs[1] / f).read_bytes()
        assert len(a) > 100 and a == b, f
    with np.load(outs[0] / "ensemble_final.npz") as za, np.load(outs[1] / "ensemble_final.npz") as zb:
        assert np.array_equal(za["positions"], zb["positions"]) and np.array_equal(za["energies"], zb["energies"])


def test_annealing_stages_sent_as_one_launch_write_the_same_files(tmp_path, monkeypatch):
    """The annealing loop of scripts/mm_run.py:171-175 (set the temperature, step 50 times, 400 times over) reaches the library as
    launches of up to AWSEM_STAGES_PER_LAUNCH stages (engine.Context._step / awsem_step_stages); report points, checkpoints and the end of
    the run cut the blocks wherever they fall.  Every file must equal, byte for byte, the run that sends one launch per stage."""
    from openawsem_amd.scripts import mm_run
    d = tmp_path / "1r69"
    shutil.copytree(data_path.data / "1r69", d, ignore=shutil.ignore_patterns("fraglib", "frags.mem"))
    shutil.copy(d / "1r69.fasta", d / "crystal_structure.fasta")
    seq = "".join(ln.strip() for ln in open(d / "1r69.fasta") if not ln.startswith(">"))
    (d / "charge.txt").write_text("".join(f"{i} {1.0 if a in 'RK' else (-1.0 if a in 'DE' else 0.0)}\n" for i, a in enumerate(seq)))
    outs = []
    for stages in ("1", "4", "16"):
        monkeypatch.setenv("AWSEM_STAGES_PER_LAUNCH", stages)
        monkeypatch.chdir(d)
        out = tmp_path / f"run_stages{stages}"
        sim = mm_run.main(["1r69", "--to", str(out), "-s", "3000", "--tempStart", "700", "--tempEnd", "300", "--replicas", "8", "--seed", "13",
                           "-f", os.path.join(os.path.dirname(mm_run.__file__), "forces_setup.py")])
        assert sim.context._max_stages == int(stages) and not sim.context._pending
        outs.append(out)
    monkeypatch.delenv("AWSEM_STAGES_PER_LAUNCH", raising=False)
    for other in outs[1:]:
        for f in ("native.pdb", "movie.dcd", "output.log", "checkpnt.chk"):
            a, b = (outs[0] / f).read_bytes(), (other / f).read_bytes()
            assert len(a) > 100 and a == b, (f, other)
        with np.load(outs[0] / "ensemble_final.npz") as za, np.load(other / "ensemble_final.npz") as zb:
            assert np.array_equal(za["positions"], zb["positions"]) and np.array_equal(za["energies"], zb["energies"])


def test_awsem_remd_runs_a_ladder_and_restarts(tmp_path, monkeypatch):
    """mm_remd.py of the reference delegates to openmmtools (absent here; its output.nc is not reproduced): outputs are
    checked for shape, label bookkeeping and restart ("parity unpinned"; the Metropolis rule itself is tested in test_ensemble.py)."""
    from openawsem_amd.scripts import mm_remd, mm_run
    src = data_path.data / "1r69"
    d = tmp_path / "1r69"
    shutil.copytree(src, d, ignore=shutil.ignore_patterns("fraglib", "frags.mem"))
    shutil.copy(d / "1r69.fasta", d / "crystal_structure.fasta")
    seq = "".join(ln.strip() for ln in open(d / "1r69.fasta") if not ln.startswith(">"))
    (d / "charge.txt").write_text("".join(f"{i} {1.0 if a in 'RK' else (-1.0 if a in 'DE' else 0.0)}\n" for i, a in enumerate(seq)))
    forces = os.path.join(os.path.dirname(mm_run.__file__), "forces_setup.py")
    monkeypatch.chdir(d)
    out = tmp_path / "remd0"
    remd = mm_remd.main(["1r69", "--to", str(out), "--replicas", "8", "--tempStart", "300", "--tempEnd", "500", "-x", "100",
                         "--iterations", "12", "--timeStep", "2", "--reportFrequency", "4", "-f", forces])
    with np.load(out / "output.npz") as z:
        assert z["energies"].shape == (12, 8) and np.isfinite(z["energies"]).all()
        assert np.allclose(z["temperatures"], np.linspace(300, 500, 8))
        assert all(sorted(row) == list(range(8)) for row in z["ladder_index"])      # a permutation of the ladder after every sweep
        # default mixing = openmmtools' swap-all: 8^3 attempts per exchange, of which those between neighbouring rungs are counted
        assert 12 * 8 ** 3 // 8 < z["attempted"].sum() < 12 * 8 ** 3 and 0 < z["accepted"].sum() <= z["attempted"].sum()
    with np.load(out / "replicas_rank0.npz") as z:
        assert z["frames"].shape == (3, 8, 368, 3) and z["positions"].shape == (8, 368, 3)
        assert int(z["iteration"]) == 12 and int(z["step_count"]) == 1200
        # hotter replicas move further from the common start
        assert np.abs(z["positions"][0] - z["positions"][1]).max() > 1e-3
    monkeypatch.chdir(d)
    remd2 = mm_remd.main(["1r69", "--to", str(tmp_path / "remd1"), "--replicas", "8", "--tempStart", "300", "--tempEnd", "500", "-x", "100",
                          "--iterations", "4", "--timeStep", "2", "--fromCheckPoint", str(out), "-f", forces])
    assert remd2.iteration == 16 and remd2.attempted.sum() > remd.attempted.sum() - 1
    # the cheaper scheme as an option: one sweep of neighbouring-rung swaps of alternating parity per exchange
    monkeypatch.chdir(d)
    remd3 = mm_remd.main(["1r69", "--to", str(tmp_path / "remd2"), "--replicas", "8", "--tempStart", "300", "--tempEnd", "500", "-x", "100",
                          "--iterations", "12", "--timeStep", "2", "--mixing", "neighbours", "-f", forces])
    assert remd3.attempted.sum() == 12 * 4 - 6 and 0 < remd3.accepted.sum() <= remd3.attempted.sum()   # 4 even pairs, 3 odd pairs


def test_awsem_run_on_the_cell_list_path(tmp_path, monkeypatch):
    """The same driver with the system forced onto the cell-list path (what a protein of more than ~390 residues takes by
    itself): device minimiser, reporters, checkpoint -- and the same physics: with identical seeds the short
    fp64 trajectories of the two paths agree."""
    from openawsem_amd.scripts import mm_run
    src = data_path.data / "1r69"
    d = tmp_path / "1r69"
    shutil.copytree(src, d, ignore=shutil.ignore_patterns("fraglib", "frags.mem"))
    shutil.copy(d / "1r69.fasta", d / "crystal_structure.fasta")
    seq = "".join(ln.strip() for ln in open(d / "1r69.fasta") if not ln.startswith(">"))
    (d / "charge.txt").write_text("".join(f"{i} {1.0 if a in 'RK' else (-1.0 if a in 'DE' else 0.0)}\n" for i, a in enumerate(seq)))
    forces = os.path.join(os.path.dirname(mm_run.__file__), "forces_setup.py")
    finals = {}
    for cell in ("1", "0"):
        monkeypatch.setenv("AWSEM_FORCE_CELL", cell)
        monkeypatch.chdir(d)
        out = tmp_path / f"run_cell{cell}"
        mm_run.main(["1r69", "--to", str(out), "-s", "400", "--tempStart", "500", "--tempEnd", "400", "--replicas", "2", "--seed", "5",
                     "-p", "Reference", "--noMinimize", "-f", forces])
        log = np.loadtxt(out / "output.log", delimiter=",", skiprows=1)
        assert log.shape == (400, 3) and np.isfinite(log).all()
        with np.load(out / "ensemble_final.npz") as z:
            finals[cell] = z["positions"].copy()
    assert np.abs(finals["1"] - finals["0"]).max() < 1e-6
    # and with the minimisation switched on (the cell path has its own device minimiser)
    monkeypatch.setenv("AWSEM_FORCE_CELL", "1")
    monkeypatch.chdir(d)
    sim = mm_run.main(["1r69", "--to", str(tmp_path / "run_min"), "-s", "400", "-m", "0", "--tempStart", "300", "--seed", "5", "-f", forces])
    assert np.isfinite(sim.context.energies()).all()
