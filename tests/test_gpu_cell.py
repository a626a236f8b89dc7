"""Cell-list path (awsem_cell.h) against the replica-resident path, the fp64 oracle and the oracle's brute-force pair
sets.  Small systems are pushed onto the cell-list path with AWSEM_FORCE_CELL=1; the synthetic multi-chain system
(BASELINE.json configs[4]) takes it by itself because it does not fit the LDS of a CU.

The per-pair arithmetic is shared by both paths and every sum is fixed point, so energies and forces of the two
paths agree to fp rounding of single contributions (the compiler may contract an a*b+c differently in the two
kernels): |dF| <= 1e-4 kJ/mol/nm in fp32, 1e-9 in fp64 -- far inside the tolerance against the oracle.  No reference fixture exists for neighbour lists or for systems of this size
("parity unpinned"): the oracle's O(N^2) scan is the checker.
"""
import contextlib
import io

import numpy as np
import pytest

import openawsem_amd as oa_pkg
from openawsem_amd import _capi, workloads
from openawsem_amd.forces import System
from oracle import awsem_oracle
from tests.common import golden_system

pytestmark = pytest.mark.gpu

def _full_system(oa):
    """All 16 terms of the reference test (con, chain, chi, excl, rama x3, contact, beta x3, pap x2, helical, fragment memory, DH)."""
    s = System(oa.system.masses.copy())
    s.virtual_sites = dict(oa.system.virtual_sites)
    s.forces = list(oa.system.forces)
    kinds = {f.kind for f in s.forces}
    assert {"contact", "excl", "beta", "pap"} <= kinds, kinds
    return s


def _ctx(system, R, precision, cell, monkeypatch):
    if cell:
        monkeypatch.setenv("AWSEM_FORCE_CELL", "1")
    else:
        monkeypatch.delenv("AWSEM_FORCE_CELL", raising=False)
    ctx = _capi.Context(_capi.load_hip_library(), system.getNumParticles(), n_replicas=R, precision=precision)
    system.upload(ctx)
    return ctx


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("protein", ["1brs", "2lzm"])
def test_cell_path_equals_resident_path(protein, precision, monkeypatch):
    oa, traj, _ = golden_system(protein)
    system = _full_system(oa)
    frames = [0, 6, 11]
    a = _ctx(system, len(frames), precision, False, monkeypatch)
    b = _ctx(system, len(frames), precision, True, monkeypatch)
    for r, f in enumerate(frames):
        a.set_positions(traj[f], replica=r)
        b.set_positions(traj[f], replica=r)
    # fp64: both paths sum 2^-32 fixed point, but the cell path meets every pair from both sides (half the energy and one
    # force each), so the two differ by rounding of the individual contributions: a few 1e-9 over ~1e4 pairs
    atol = 1e-4 if precision == 0 else 1e-7
    ea, eb = a.energy_by_group(), b.energy_by_group()
    assert np.abs(ea).sum() > 0 and np.abs(ea - eb).max() <= atol, np.abs(ea - eb).max()
    for redistribute in (False, True):
        fa, fb = a.forces(redistribute=redistribute), b.forces(redistribute=redistribute)
        # fp32: the resident path books forces as Q19.12 (2.4e-4 kJ/mol/nm per bit, x/y packed in one LDS word), the cell
        # path as Q31.32 -> agreement to a few bits of the coarser format, i.e. ~1e-5 of the largest force
        ftol = atol if precision == 1 else 1e-5 * np.abs(fb).max() + 1e-3
        assert np.abs(fa - fb).max() <= ftol, np.abs(fa - fb).max()
    # single group masks take the same route
    assert np.abs(a.energy(1 << 22) - b.energy(1 << 22)).max() <= atol
    # the cell path itself is deterministic although its list order is not
    assert np.array_equal(fb, b.forces(redistribute=True)) and np.array_equal(eb, b.energy_by_group())
    # and against the fp64 oracle
    cpu = awsem_oracle.make_context(system, n_replicas=len(frames), threads=3)
    for r, f in enumerate(frames):
        cpu.set_positions(traj[f], replica=r)
    ec, fc = cpu.energy_by_group(), cpu.forces(redistribute=True)
    tol_e, tol_f = (1e-4, 3e-3) if precision == 0 else (1e-7, 1e-7)   # energies are 2^-32 fixed-point sums
    assert (np.abs(eb - ec) / np.maximum(np.abs(ec), 1.0)).max() <= tol_e
    assert np.linalg.norm(fb - fc) <= tol_f * np.linalg.norm(fc)


@pytest.mark.parametrize("precision", [0, 1])
def test_cell_neighbour_lists_equal_brute_force(precision, monkeypatch):
    oa, traj, _ = golden_system("1brs")
    system = _full_system(oa)
    gpu = _ctx(system, 2, precision, True, monkeypatch)
    cpu = _capi.Context(awsem_oracle.load(), oa.natoms, 2, precision=precision)
    system.upload(cpu)
    for ctx in (gpu, cpu):
        ctx.set_positions(traj[1], replica=0)
        ctx.set_positions(traj[9], replica=1)
    for kind in (_capi.LIST_EXCL, _capi.LIST_CONTACT):
        for replica in (0, 1):
            for skin in (0.0, 0.15, 0.4):
                a, b = gpu.neighbors(kind, 0, replica, skin), cpu.neighbors(kind, 0, replica, skin)
                assert len(b) > 20 and np.array_equal(a, b), f"kind {kind} replica {replica} skin {skin}: {len(a)} vs {len(b)}"
    for kind, count in ((_capi.LIST_BETA, 3), (_capi.LIST_PAP, 2)):      # (acceptor, donor) pairs with a non-zero table entry
        for index in range(count):
            for skin in (0.0, 0.2):
                a, b = gpu.neighbors(kind, index, 1, skin), cpu.neighbors(kind, index, 1, skin)
                assert len(b) > 20 and np.array_equal(a, b), f"kind {kind} index {index} skin {skin}: {len(a)} vs {len(b)}"
    # an evaluation after a neighbour query rebuilds the lists with the standard cutoff
    e = gpu.energy_by_group()
    assert np.isfinite(e).all()


def test_cell_dynamics_tracks_resident_path(monkeypatch):
    oa, traj, _ = golden_system("1ubq")
    system = _full_system(oa)
    T = np.array([300.0, 700.0])
    out = []
    for cell in (False, True):
        ctx = _ctx(system, 2, 1, cell, monkeypatch)     # fp64: differences are summation-order noise only
        ctx.set_positions(traj[2], replica=-1)
        ctx.set_velocities_to_temperature(T, 4)
        for chunk in (30, 270):      # long enough for several list rebuilds at 700 K
            ctx.step(chunk, T, 0.002, 1.0, 21)
        ctx.synchronize()
        out.append((np.stack([ctx.get_positions(r) for r in range(2)]), np.stack([ctx.get_velocities(r) for r in range(2)])))
    assert np.abs(out[0][0][1] - traj[2]).max() > 0.05
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-8 and np.abs(out[0][1] - out[1][1]).max() < 1e-6


def test_hydrogen_bond_terms_need_the_excluded_volume_term_on_the_cell_path(monkeypatch):
    """The rebuild test of the cell path watches the members of the contact / excluded-volume lists (CA, CB, O)."""
    oa, _, _ = golden_system("1ubq")
    s = System(oa.system.masses.copy())
    s.virtual_sites = dict(oa.system.virtual_sites)
    s.forces = [f for f in oa.system.forces if f.kind != "excl"]
    monkeypatch.setenv("AWSEM_FORCE_CELL", "1")
    ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, 1)
    with pytest.raises(_capi.AwsemError, match="excluded-volume term"):
        s.upload(ctx)


@pytest.mark.parametrize("protein", ["1ubq", "1brs"])
def test_cell_path_reproduces_the_golden_energies_with_all_terms(protein, monkeypatch):
    """The reference's own fixture (tests/data/<p>_energies.csv) through the cell-list path in fp64 mode."""
    from tests.common import COLUMNS, FORCE_GROUP
    oa, traj, golden = golden_system(protein)
    ctx = _ctx(_full_system(oa), len(traj), 1, True, monkeypatch)
    for r in range(len(traj)):
        ctx.set_positions(traj[r], replica=r)
    e = ctx.energy_by_group()
    for c in COLUMNS:
        assert np.abs(e[:, FORCE_GROUP[c]] - golden[c]).max() <= 1e-5, c


def test_synthetic_multichain_system_takes_the_cell_path():
    """~2600 residues (16 chains) with the hydrogen-bond terms: too large for LDS, small enough for the O(N^2) oracle in seconds."""
    with contextlib.redirect_stdout(io.StringIO()):
        oa = workloads.build_synthetic(n_chains=16, seed=3, hbonds=True)
    assert oa.nres == 16 * 164
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, 1, precision=0)
    oa.system.upload(gpu)
    cpu = _capi.Context(awsem_oracle.load(), oa.natoms, 1, precision=0)
    oa.system.upload(cpu)
    cpu._call("set_threads", 8)
    for ctx in (gpu, cpu):
        ctx.set_positions(oa.pdb.positions)
    for kind in (_capi.LIST_EXCL, _capi.LIST_CONTACT):
        a, b = gpu.neighbors(kind, 0, 0, 0.15), cpu.neighbors(kind, 0, 0, 0.15)
        assert len(b) > 1000 and np.array_equal(a, b)
    for kind, index in ((_capi.LIST_BETA, 1), (_capi.LIST_PAP, 0)):
        a, b = gpu.neighbors(kind, index, 0, 0.2), cpu.neighbors(kind, index, 0, 0.2)
        assert len(b) > 1000 and np.array_equal(a, b)
    eg, ec = gpu.energy_by_group(), cpu.energy_by_group()
    assert abs(ec[0, 27]) > 100 and abs(ec[0, 28]) > 10      # beta and P-AP are at work
    assert (np.abs(eg - ec) / np.maximum(np.abs(ec), 1.0)).max() <= 1e-4
    fg, fc = gpu.forces(), cpu.forces()
    assert np.linalg.norm(fg - fc) <= 1e-3 * np.linalg.norm(fc)
    # dynamics: lists are rebuilt on the device as the chains move; energies stay finite and change
    gpu.set_velocities_to_temperature([300.0], 1)
    gpu.step(200, [300.0], 0.002, 1.0, 1)
    e2 = gpu.energy_by_group()
    assert np.isfinite(e2).all() and not np.array_equal(e2, eg)


def test_cell_path_minimiser_follows_the_resident_one(monkeypatch):
    """awsem_minimize on the cell-list path is the resident kernel's FIRE iteration spread over grid-wide kernels: in fp64 the
    first 25 iterations agree to summation noise, and a full minimisation reaches the tolerance in a minimum of the same depth."""
    oa, traj, _ = golden_system("1ubq")
    system = _full_system(oa)
    res = {}
    for cell in (False, True):
        ctx = _ctx(system, 2, 1, cell, monkeypatch)
        ctx.set_positions(traj[3], replica=0)
        ctx.set_positions(traj[9], replica=1)
        ctx.compute_virtual_sites()
        ctx.minimize(1e-6, 25)
        x25 = ctx.get_state(velocities=False)[0]
        it, rms = ctx.minimize(10.0, 20000)
        res[cell] = (x25, it, rms, ctx.energy(), ctx.get_state(velocities=False)[0])
    assert np.abs(res[True][0] - res[False][0]).max() < 1e-7
    assert np.all(res[True][2] <= 10.0) and np.all(res[True][1] < 20000)
    assert np.all(np.abs(res[True][3] - res[False][3]) < 0.02 * np.abs(res[False][3])), (res[True][3], res[False][3])
    # the synthetic multi-chain system is what the path exists for: all replicas relaxed on the device
    with contextlib.redirect_stdout(io.StringIO()):
        big = workloads.build_synthetic(n_chains=4, seed=5, hbonds=True)
    monkeypatch.setenv("AWSEM_FORCE_CELL", "1")
    ctx = _capi.Context(_capi.load_hip_library(), big.natoms, 2, precision=0)
    big.system.upload(ctx)
    ctx.set_positions(big.pdb.positions)
    e0 = ctx.energy()
    it, rms = ctx.minimize(10.0, 5000)
    assert np.all(rms <= 10.0) and np.all(ctx.energy() < e0), (it, rms)


def test_row_overflow_during_dynamics_is_replayed_not_fatal(monkeypatch, capfd):
    """Neighbour rows that are too short for the structure (here: deliberately sized to 40 % of what the start structure needs, the
    stand-in for a chain that compacts during annealing) overflow at a rebuild inside an asynchronous launch.  The launches since
    the last synchronisation are then replayed from a snapshot with longer rows.  The random stream is counter
    based and every sum is fixed point, so the replayed trajectory must be BIT-IDENTICAL to a run whose rows were long enough
    from the start -- in production precision -- and an energy query after asynchronous steps must trigger the repair instead of
    swallowing the overflow."""
    oa, traj, _ = golden_system("1ubq")
    system = _full_system(oa)
    x0 = traj[5]
    T = np.array([300.0, 400.0])
    out = {}
    for rows in ("roomy", "tight"):
        monkeypatch.setenv("AWSEM_VERBOSE", "1")
        monkeypatch.setenv("AWSEM_CELL_NEIGHBORS", "512" if rows == "roomy" else "16")
        monkeypatch.setenv("AWSEM_CELL_HEADROOM", "2.0" if rows == "roomy" else "0.4")
        ctx = _ctx(system, 2, 0, True, monkeypatch)
        ctx.set_positions(x0, replica=-1)
        ctx.compute_virtual_sites()
        ctx.set_velocities_to_temperature(T, 4)
        for chunk in (40, 40, 120):          # asynchronous launches, resolved by the energy query
            ctx.step(chunk, T, 0.002, 1.0, 21)
        e = ctx.energy()
        ctx.step(100, T, 0.002, 1.0, 21)
        ctx.synchronize()
        out[rows] = (ctx.get_state(), e, capfd.readouterr().err)
        ctx.close()
    assert "rows regrown" not in out["roomy"][2]
    assert "rows regrown" in out["tight"][2] and "replaying" in out["tight"][2], out["tight"][2]
    (xa, va), (xb, vb) = out["roomy"][0], out["tight"][0]
    assert np.isfinite(xa).all() and np.abs(xa[0] - x0).max() > 0.05
    assert np.array_equal(xa, xb) and np.array_equal(va, vb)
    assert np.array_equal(out["roomy"][1], out["tight"][1])


def test_full_size_synthetic_system_rows_energies_forces_and_dynamics():
    """BASELINE.json configs[4] at its full size: 122 x 2lzm = 20 008 residues, 117 974 atoms, con/chain/chi/excl/rama/contact+burial.
    Rows of both binnings against the oracle's brute-force O(N^2) scan (2e8 and 9e8 pair tests, 8 threads), energies and forces
    against the oracle, then 200 steps that take the system through at least one list rebuild."""
    with contextlib.redirect_stdout(io.StringIO()):
        oa = workloads.build_synthetic()
    assert oa.nres == 20008
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, 1, precision=0)
    oa.system.upload(gpu)
    cpu = _capi.Context(awsem_oracle.load(), oa.natoms, 1, precision=0)
    oa.system.upload(cpu)
    cpu._call("set_threads", 8)
    for ctx in (gpu, cpu):
        ctx.set_positions(oa.pdb.positions)
    for kind in (_capi.LIST_EXCL, _capi.LIST_CONTACT):
        a, b = gpu.neighbors(kind, 0, 0, 0.15, capacity=1 << 19), cpu.neighbors(kind, 0, 0, 0.15, capacity=1 << 19)
        assert len(b) > 50000 and np.array_equal(a, b)
    eg, ec = gpu.energy_by_group(), cpu.energy_by_group()
    assert np.abs(ec[0, 20:23]).min() > 1e4
    assert (np.abs(eg - ec) / np.maximum(np.abs(ec), 1.0)).max() <= 1e-4
    fg, fc = gpu.forces(), cpu.forces()
    assert np.linalg.norm(fg - fc) <= 1e-3 * np.linalg.norm(fc)
    assert np.abs(fg - fc).max() <= 1e-3 * np.abs(fc).max()
    # dynamics through a rebuild: at 600 K a CA moves half the default skin (0.1 nm) within ~100 steps
    gpu.set_velocities_to_temperature([600.0], 1)
    x0 = gpu.get_positions(0)
    gpu.step(200, [600.0], 0.002, 1.0, 1)
    gpu.synchronize()
    x1 = gpu.get_positions(0)
    moved = np.linalg.norm((x1 - x0)[np.array(oa.ca)], axis=1).max()
    assert moved > 0.12, moved               # further than skin / 2: the lists were rebuilt on the device
    e2 = gpu.energy_by_group()
    assert np.isfinite(e2).all() and np.isfinite(x1).all() and not np.array_equal(e2, eg)
    # the rows after the run still equal the brute force on the new coordinates
    cpu.set_positions(x1)
    a, b = gpu.neighbors(_capi.LIST_CONTACT, 0, 0, 0.15, capacity=1 << 19), cpu.neighbors(_capi.LIST_CONTACT, 0, 0, 0.15, capacity=1 << 19)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("precision", [0, 1])
def test_five_launch_step_equals_the_eleven_launch_step_bit_for_bit(precision, monkeypatch):
    """Round 3: a cell-list step is window kernel (bonded terms, places the virtual sites) -> ONE rebuild launch (grid barriers of its
    own, leaves at once on most steps) -> density + excluded volume -> pairs -> chain rule -> hydrogen bonds -> integrator (which also
    leaves the next step's member records, displacement flag and bounding boxes); rounds 1 and 2 ran pre / geometry / count / scan /
    fill / rows / hydrogen-bond rows as launches of their own (AWSEM_CELL_FUSED=0).  Counter-based random numbers and fixed-point sums:
    both must produce the SAME trajectory, bit for bit, through list rebuilds (700 K), graph replays (>= 10 steps) and single steps,
    with all terms (hydrogen bonds included) and with centre-of-mass removal."""
    oa, traj, _ = golden_system("1brs")
    system = _full_system(oa)
    T = np.array([300.0, 700.0, 500.0])
    out = []
    for fused in ("1", "0"):
        monkeypatch.setenv("AWSEM_CELL_FUSED", fused)
        ctx = _ctx(system, 3, precision, True, monkeypatch)
        ctx.set_positions(traj[2], replica=-1)
        ctx.set_velocities_to_temperature(T, 4)
        log = []
        for chunk in (7, 30, 123):
            ctx.step(chunk, T, 0.002, 1.0, 21)
            log.append(ctx.energy_by_group())
        ctx.synchronize()
        out.append((ctx.get_state(), log))
        ctx.close()
    monkeypatch.delenv("AWSEM_CELL_FUSED", raising=False)
    assert np.array_equal(out[0][0][0], out[1][0][0]) and np.array_equal(out[0][0][1], out[1][0][1])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)
    assert np.abs(out[0][0][0][1] - traj[2]).max() > 0.05
