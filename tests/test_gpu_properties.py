"""Size-independent properties of the HIP path (no oracle needed) and edge cases of the C ABI:
exact momentum balance of the fixed-point pair forces, asynchronous launches, empty / tiny systems, bench-size ensembles."""
import numpy as np
import pytest

from openawsem_amd import _capi, workloads
from openawsem_amd.forces import HarmonicBondForce, System
from tests.common import golden_system

pytestmark = pytest.mark.gpu
FP32, FP64 = 0, 1


def _ctx(system, natoms, R=1, precision=FP32):
    ctx = _capi.Context(_capi.load_hip_library(), natoms, n_replicas=R, precision=precision)
    system.upload(ctx)
    return ctx


@pytest.mark.parametrize("precision", [FP32, FP64])
def test_pair_terms_obey_newtons_third_law_exactly(precision):
    """A pair term books +F on one atom and the bit-identical -F on the other in fixed point, so the raw (not
    redistributed) forces of the pair-only groups sum to exactly zero, component by component, whatever the order the
    lists were walked in: fragment memory (23), beta (27), P-AP (28), helical (29), Debye-Hueckel (30)."""
    oa, traj, _ = golden_system("1brs")
    gpu = _ctx(oa.system, oa.natoms, R=3, precision=precision)
    for r, f in enumerate((0, 5, 11)):
        gpu.set_positions(traj[f], replica=r)
    for group in (23, 27, 28, 29, 30):
        f = gpu.forces(1 << group, redistribute=False)
        assert np.abs(f).max() > 0, group
        # the sums are exact in fixed point; in double the 1e3 terms of up to 1e3 kJ/mol/nm leave at most rounding of the readout
        assert np.all(f.sum(axis=1) == 0.0), (group, f.sum(axis=1))


def test_asynchronous_launches_equal_synchronised_ones():
    """awsem_step returns before the kernel has run; 40 launches in a row (more than the 32-deep ring) with changing
    temperatures give the same trajectory as launches that are synchronised one by one."""
    oa, traj, _ = golden_system("1ubq")
    out = []
    for sync in (False, True):
        gpu = _ctx(oa.system, oa.natoms, R=2)
        gpu.set_positions(traj[0])
        gpu.compute_virtual_sites()
        gpu.set_velocities_to_temperature([300.0, 300.0], 5)
        for k in range(40):
            T = np.array([300.0 + k, 350.0 - k])
            gpu.step(3, T, 0.002, 1.0, 5)
            T[:] = -1.0            # the library has copied the temperatures: the caller may reuse its buffer at once
            if sync:
                gpu.synchronize()
        out.append((gpu.get_positions(0), gpu.get_positions(1), gpu.get_velocities(1)))
        assert gpu.step_count == 120
    for a, b in zip(*out):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("lean", ["1", "0"])
def test_stages_in_one_launch_equal_one_launch_per_stage(lean, monkeypatch):
    """awsem_step_stages -- the annealing loop's stages in one launch, 16 at most -- gives bit for bit the trajectory of one awsem_step
    per stage: 19 stages (a launch of 16 and one of 3) of 7 steps with per-replica temperatures, on both instantiations of the step
    kernel; the step counter advances alike, and the caller's temperature block is free for reuse on return."""
    monkeypatch.setenv("AWSEM_STEP_LEAN", lean)
    oa, traj, _ = golden_system("1ubq")
    T = np.stack([np.array([800.0 - 20.0 * g, 300.0 + 11.0 * g, 450.0]) for g in range(19)])
    out = []
    for staged in (False, True):
        gpu = _ctx(oa.system, oa.natoms, R=3)
        gpu.set_positions(traj[0])
        gpu.compute_virtual_sites()
        gpu.set_velocities_to_temperature([800.0, 300.0, 450.0], 11)
        gpu.step(5, T[0], 0.002, 1.0, 11)           # lists in place, a step counter that is not zero
        if staged:
            block = T.copy()
            gpu.step_stages(7, block, 0.002, 1.0, 11)
            block[:] = -1.0
        else:
            for row in T:
                gpu.step(7, row, 0.002, 1.0, 11)
        assert gpu.step_count == 5 + 19 * 7
        out.append([gpu.get_positions(r) for r in range(3)] + [gpu.get_velocities(r) for r in range(3)])
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    assert not np.array_equal(out[0][0], out[0][1])   # the replicas did run at their own temperatures


@pytest.mark.parametrize("mode", ["fp64", "cell"])
def test_stages_on_the_other_paths_and_their_edge_cases(mode, monkeypatch):
    """awsem_step_stages in the fp64 check mode (its own step kernel) and on the cell-list path (which runs the stages one by one): the
    trajectory of one awsem_step per stage; zero stages and zero steps per stage advance nothing; bad arguments are refused."""
    if mode == "cell":
        monkeypatch.setenv("AWSEM_FORCE_CELL", "1")
    oa, traj, _ = golden_system("1ubq")
    T = np.stack([np.array([600.0 - 30.0 * g, 320.0]) for g in range(5)])
    out = []
    for staged in (False, True):
        gpu = _ctx(oa.system, oa.natoms, R=2, precision=FP64 if mode == "fp64" else FP32)
        gpu.set_positions(traj[0])
        gpu.compute_virtual_sites()
        gpu.set_velocities_to_temperature([600.0, 320.0], 3)
        if staged:
            gpu.step_stages(6, T, 0.002, 1.0, 4)
            before = gpu.step_count
            gpu.step_stages(6, T[:0], 0.002, 1.0, 4)      # no stage
            gpu.step_stages(0, T, 0.002, 1.0, 4)          # stages of no steps
            assert gpu.step_count == before
            with pytest.raises(_capi.AwsemError, match="bad arguments"):
                gpu.step_stages(6, T, -0.002, 1.0, 4)
            with pytest.raises(_capi.AwsemError, match="bad arguments"):
                gpu.step_stages(-1, T, 0.002, 1.0, 4)
        else:
            for row in T:
                gpu.step(6, row, 0.002, 1.0, 4)
        assert gpu.step_count == 30
        out.append((gpu.get_positions(0), gpu.get_positions(1), gpu.get_velocities(0), gpu.get_velocities(1)))
    for a, b in zip(*out):
        assert np.array_equal(a, b)


def test_empty_and_tiny_systems():
    """Two bonded particles and a spectator, an empty bond list next to it: energies, forces and dynamics behave."""
    system = System(np.array([12.0, 12.0, 12.0]))
    bond = HarmonicBondForce()
    bond.addBond(0, 1, 0.38, 1000.0)
    bond.setForceGroup(20)
    empty = HarmonicBondForce()
    empty.setForceGroup(21)
    system.addForce(bond)
    system.addForce(empty)
    for precision in (FP32, FP64):
        gpu = _ctx(system, 3, R=2, precision=precision)
        x = np.array([[0.0, 0.0, 0.0], [0.5, 0.0, 0.0], [3.0, 3.0, 3.0]])
        gpu.set_positions(x)
        e = gpu.energy_by_group()
        assert np.allclose(e[:, 20], 0.5 * 1000.0 * 0.12 ** 2, rtol=1e-6) and np.all(e[:, 21] == 0)
        f = gpu.forces()
        assert np.allclose(f[0, 0], [120.0, 0, 0], atol=1e-3) and np.allclose(f[0, 1], [-120.0, 0, 0], atol=1e-3) and np.all(f[0, 2] == 0)
        gpu.set_velocities(np.zeros((3, 3)))
        gpu.step(200, [0.0, 0.0], 0.002, 5.0, 1)      # T = 0: pure damped relaxation towards r0
        y = gpu.get_positions(1)
        assert abs(np.linalg.norm(y[1] - y[0]) - 0.38) < 0.02 and np.allclose(y[2], x[2])
        it, rms = gpu.minimize(1e-3, 5000)
        z = gpu.get_positions(0)
        assert np.all(rms <= 1e-3) and abs(np.linalg.norm(z[1] - z[0]) - 0.38) < 1e-4


def test_bench_size_ensemble_stays_finite_and_reproducible():
    """BASELINE.json's full size (1024 replicas of 1r69): two identical runs agree bit for bit, replicas with different
    random streams differ, all energies stay finite."""
    oa = workloads.build_1r69()
    R = 1024
    runs = []
    for _ in range(2):
        gpu = _ctx(oa.system, oa.natoms, R=R)
        gpu.set_positions(oa.pdb.positions)
        gpu.set_velocities_to_temperature(np.full(R, 800.0), 2026)
        for stage in range(3):
            gpu.step(50, np.full(R, 800.0 - 1.5 * stage), 0.002, 1.0, 2026)
        e = gpu.energy()
        assert np.isfinite(e).all()
        runs.append((e, gpu.get_positions(0), gpu.get_positions(R - 1)))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert np.abs(runs[0][1] - runs[0][2]).max() > 1e-3
    assert len(np.unique(runs[0][0])) > R // 2


def test_bulk_state_and_device_kinetic_energy_match_the_per_replica_calls():
    oa, traj, _ = golden_system("1ubq")
    R = 5
    gpu = _ctx(oa.system, oa.natoms, R=R)
    for r in range(R):
        gpu.set_positions(traj[r], replica=r)
    gpu.set_velocities_to_temperature(np.linspace(200.0, 600.0, R), 11)
    gpu.step(7, np.linspace(200.0, 600.0, R), 0.002, 1.0, 11)
    x, v = gpu.get_state()
    for r in range(R):
        assert np.array_equal(x[r], gpu.get_positions(r)) and np.array_equal(v[r], gpu.get_velocities(r))
    m = oa.system.masses
    ke = np.array([0.5 * (m[:, None] * v[r] ** 2).sum() for r in range(R)])
    assert np.allclose(gpu.kinetic_energy(), ke, rtol=1e-6)
    dof = 3 * (m > 0).sum()
    assert np.all(np.abs(2 * ke / (dof * 0.00831446261815324) / np.linspace(200.0, 600.0, R) - 1) < 0.6)    # kinetic temperature in the range of the bath (666 degrees of freedom, 7 steps after the draw)
    # round trip through set_state
    other = _ctx(oa.system, oa.natoms, R=R)
    other.set_state(x, v)
    y, w = other.get_state()
    assert np.array_equal(x, y) and np.array_equal(v, w)
    assert np.array_equal(other.energy(), gpu.energy())


def test_annealing_folds_1r69_towards_its_native_structure():
    """End-to-end physics: the default force field with the single-memory fragment term of the native structure, annealed
    800 -> 200 K, must fold the 434 repressor domain -- Q against the crystal structure rises and the potential energy
    falls.  (tools/anneal_q.py is the long version: 1024 replicas x 10^6 steps in 40 s, <Q> 0.26 -> 0.67, 99 % above 0.5.)
    No reference fixture exists for a trajectory ("parity unpinned"); a force with a wrong sign anywhere fails this."""
    import contextlib
    import io
    import os
    from openawsem_amd import workloads
    from openawsem_amd.functionTerms import biasTerms
    from tests.common import GOLDEN
    R, per_stage = 128, 500
    with contextlib.redirect_stdout(io.StringIO()):
        oa = workloads.build_1r69()
        oa.addForcesWithDefaultForceGroup([biasTerms.q_value(oa, os.path.join(GOLDEN, "1r69.pdb"), forceGroup=1)])
    ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=R)
    oa.system.upload(ctx)
    ctx.set_positions(oa.pdb.positions)
    ctx.minimize(10.0, 0)
    q0, e0 = ctx.energy(1 << 1).mean(), ctx.energy().mean()
    ctx.set_velocities_to_temperature(np.full(R, 800.0), 7)
    for i in range(400):
        ctx.step(per_stage, np.full(R, 800.0 - 1.5 * i), 0.002, 1.0, 7)
    q, e = ctx.energy(1 << 1), ctx.energy()
    assert q0 < 0.2 and q.mean() > 0.45 and (q > 0.4).mean() > 0.7, (q0, q.mean())
    assert e.mean() < e0 - 500.0, (e0, e.mean())


def test_exploding_forces_are_reported_not_wrapped(monkeypatch):
    """Two bonded atoms placed 40 nm from their rest length: 2e6 kJ/mol/nm, beyond the Q19.12 range of the fp32 accumulators
    of the replica-resident path (the cell-list path accumulates in Q31.32 and has no such limit)."""
    monkeypatch.setenv("AWSEM_FORCE_CELL", "0")
    system = System(np.array([12.0, 12.0]))
    f = HarmonicBondForce(group=20)
    f.addBond(0, 1, 0.38, 50208.0)
    system.addForce(f)
    ctx = _capi.Context(_capi.load_hip_library(), 2, n_replicas=1, precision=0)
    system.upload(ctx)
    ctx.set_positions(np.array([[0.0, 0.0, 0.0], [40.0, 0.0, 0.0]]))
    ctx.set_velocities_to_temperature([300.0], 1)
    with pytest.raises(_capi.AwsemError, match="fixed-point accumulators"):
        ctx.step(1, [300.0], 0.002, 1.0, 1)
        ctx.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("cell", [False, True])
def test_snapshot_returns_what_the_synchronous_queries_return(cell, monkeypatch):
    """awsem_snapshot_begin / _end (include/awsem_hip.h): the asynchronous reads behind the reporters of scripts/mm_run.py:157-164.  A
    snapshot is taken BEHIND the launches already queued and is not disturbed by launches queued after it: energies (all replicas, or
    one replica alone with the others NaN), kinetic energies, one replica's positions and the ensemble state must equal, bit for bit,
    what awsem_energy / awsem_kinetic_energy / awsem_get_positions / awsem_get_state return at the same point -- on both kernel paths,
    with both slots in flight, and a slot cannot be begun twice."""
    if cell:
        monkeypatch.setenv("AWSEM_FORCE_CELL", "1")
    else:
        monkeypatch.delenv("AWSEM_FORCE_CELL", raising=False)
    oa, traj, _ = golden_system("1ubq")
    R = 4
    T = np.linspace(300.0, 600.0, R)

    def fresh():
        ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=R, precision=0)
        oa.system.upload(ctx)
        for r in range(R):
            ctx.set_positions(traj[r], replica=r)
        ctx.set_velocities_to_temperature(T, 3)
        ctx.step(25, T, 0.002, 1.0, 9)
        return ctx

    # reference: synchronous queries after 25 and after 50 steps
    ctx = fresh()
    want = []
    for _ in range(2):
        want.append(dict(e=ctx.energy(), e22=ctx.energy(1 << 22), ke=ctx.kinetic_energy(), x2=ctx.get_positions(2), state=ctx.get_state()))
        ctx.step(25, T, 0.002, 1.0, 9)
    ctx.close()

    ctx = fresh()
    ctx.snapshot_begin(0, 0xFFFFFFFF, -1, 2, True)           # after 25 steps: all replicas' energies, replica 2, the state
    with pytest.raises(_capi.AwsemError):
        ctx.snapshot_begin(0)                                 # the slot is taken
    ctx.step(25, T, 0.002, 1.0, 9)                            # queued behind the snapshot
    ctx.snapshot_begin(1, 1 << 22, 1, 2, False)               # after 50 steps: the contact energy of replica 1 alone
    ctx.step(25, T, 0.002, 1.0, 9)
    a = ctx.snapshot_end(0, energy=True, replica_positions=True, state=True)
    b = ctx.snapshot_end(1, energy=True, replica_positions=True, state=False)
    with pytest.raises(_capi.AwsemError):
        ctx.snapshot_end(1)                                   # nothing pending there any more
    assert np.array_equal(a["energy"], want[0]["e"]) and np.array_equal(a["kinetic"], want[0]["ke"])
    assert np.array_equal(a["positions"], want[0]["x2"])
    assert np.array_equal(a["state"][0], want[0]["state"][0]) and np.array_equal(a["state"][1], want[0]["state"][1])
    assert b["energy"][1] == want[1]["e22"][1] and np.isnan(np.delete(b["energy"], 1)).all() if not cell else np.array_equal(b["energy"], want[1]["e22"])
    assert np.array_equal(b["kinetic"], want[1]["ke"]) and np.array_equal(b["positions"], want[1]["x2"])
    ctx.close()
