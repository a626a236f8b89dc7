"""GPU parity: libawsem_hip.so (through the C ABI) against the golden energies of the reference and the fp64 CPU
oracle, on the reference's own test inputs (5 proteins x 12 trajectory frames, all 16 test forces).

Tolerances (BASELINE.json north_star): fp32 kernels -- energies <= 1e-4 relative, forces <= 1e-3 relative;
fp64 check mode -- the reference's own atol = 1e-5 kJ/mol on the golden columns.
"""
import numpy as np
import pytest

from openawsem_amd import _capi
from oracle import awsem_oracle
from tests.common import COLUMNS, FORCE_GROUP, PROTEINS, energies_by_column, golden_system

pytestmark = pytest.mark.gpu

FP32, FP64 = 0, 1


def _gpu(oa, n_replicas, precision):
    ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=n_replicas, device=0, precision=precision)
    oa.system.upload(ctx)
    return ctx


@pytest.mark.parametrize("protein", PROTEINS)
def test_fp64_mode_matches_reference_golden_energies(protein):
    oa, traj, golden = golden_system(protein)
    got = energies_by_column(_gpu(oa, len(traj), FP64), traj)
    for column in COLUMNS:
        assert np.allclose(got[column], golden[column], atol=1e-5, rtol=0), \
            f"{protein} {column}: max |diff| = {np.abs(got[column] - golden[column]).max():.3e}"


@pytest.mark.parametrize("protein", PROTEINS)
def test_fp32_energies_within_1e4_relative_of_golden(protein):
    oa, traj, golden = golden_system(protein)
    got = energies_by_column(_gpu(oa, len(traj), FP32), traj)
    for column in COLUMNS:
        # relative to the column's magnitude; Backbone sums large cancelling spring energies -> small absolute floor
        scale = np.maximum(np.abs(golden[column]), 1.0)
        rel = np.abs(got[column] - golden[column]) / scale
        assert rel.max() <= 1e-4, f"{protein} {column}: max rel diff {rel.max():.3e}"


def _fm_bin_edge_allowance(oa, x):
    """The fragment-memory potential is piecewise linear in r (reference templateTerms.py:273-281), so its force
    jumps at every table bin edge.  A bond whose length sits within fp32 resolution of an edge (frame 0 of each
    golden trajectory comes from 3-decimal PDB coordinates, where this happens) may legitimately take the slope of
    either bin; allow exactly that jump for such bonds."""
    fm = [f for f in oa.system.forces if f.kind == "fragment_memory"][0]
    i, j = fm.pairs[:, 0], fm.pairs[:, 1]
    r = np.linalg.norm(x[i] - x[j], axis=1)
    q = r / fm.dr
    near = np.abs(q - np.round(q)) < 2e-4
    idx = np.clip(np.round(q[near]).astype(int), 1, fm.table.shape[1] - 2)
    rows = fm.table[near]
    ar = np.arange(len(idx))
    jump = np.abs((rows[ar, idx + 1] - rows[ar, idx]) - (rows[ar, idx] - rows[ar, idx - 1])) / fm.dr
    return float(np.sqrt(2.0) * fm.k_fm * jump.sum())


@pytest.mark.parametrize("protein", PROTEINS)
@pytest.mark.parametrize("precision,tol", [(FP32, 1e-3), (FP64, 1e-7)])
def test_forces_match_oracle_per_group(protein, precision, tol):
    oa, traj, _ = golden_system(protein)
    frames = [0, 5, 11]
    gpu = _gpu(oa, len(frames), precision)
    cpu = awsem_oracle.make_context(oa.system, n_replicas=len(frames), threads=3)
    for r, f in enumerate(frames):
        gpu.set_positions(traj[f], replica=r)
        cpu.set_positions(traj[f], replica=r)
    for name, g in FORCE_GROUP.items():
        if name in ("Membrane", "ER", "TBM_Q"):
            continue
        fg, fc = gpu.forces(1 << g, redistribute=False), cpu.forces(1 << g, redistribute=False)
        for r in range(len(frames)):
            err, ref = np.linalg.norm(fg[r] - fc[r]), np.linalg.norm(fc[r])
            # absolute floor: all-helical frames have P-AP force norms of 1e-8 kJ/mol/nm (fp32 denormal products)
            allowed = tol * ref + 1e-4
            if name == "Fragment" and precision == FP32:
                allowed += _fm_bin_edge_allowance(oa, traj[frames[r]])
            assert err <= allowed, f"{protein} {name} frame {frames[r]}: force error {err:.3e} vs |F| {ref:.3e}"
    # total, after virtual-site redistribution (what the integrator consumes)
    fg, fc = gpu.forces(redistribute=True), cpu.forces(redistribute=True)
    allowed = tol * np.linalg.norm(fc)
    if precision == FP32:
        allowed += sum(_fm_bin_edge_allowance(oa, traj[f]) for f in frames)
    assert np.linalg.norm(fg - fc) <= allowed, f"{protein} total: {np.linalg.norm(fg - fc):.3e}"
    virtual = sorted(oa.system.virtual_sites)
    assert np.all(fg[:, virtual] == 0)


def test_forces_are_bitwise_reproducible():
    oa, traj, _ = golden_system("1brs")
    gpu = _gpu(oa, 12, FP32)
    for r in range(12):
        gpu.set_positions(traj[r], replica=r)
    a = gpu.forces()
    for _ in range(3):
        assert np.array_equal(a, gpu.forces())
    e = gpu.energy_by_group()
    assert np.array_equal(e, gpu.energy_by_group())


def test_same_frame_in_every_replica_gives_identical_results():
    oa, traj, _ = golden_system("1ubq")
    gpu = _gpu(oa, 300, FP32)          # more workgroups than CUs
    gpu.set_positions(traj[4], replica=-1)
    f = gpu.forces()
    assert np.all(f == f[0])
    e = gpu.energy_by_group()
    assert np.all(e == e[0])


@pytest.mark.parametrize("precision,tol", [(FP32, 2e-5), (FP64, 1e-10)])
def test_langevin_steps_track_oracle_with_shared_rng(precision, tol):
    """OpenMM's LangevinIntegrator has no reference fixture (parity unpinned): the GPU integrator is compared
    step-locked with the oracle's restatement using the shared Philox stream."""
    oa, traj, _ = golden_system("1ubq")
    R = 3
    gpu = _gpu(oa, R, precision)
    cpu = awsem_oracle.make_context(oa.system, n_replicas=R, threads=3)
    T = np.array([300.0, 500.0, 800.0])
    for ctx in (gpu, cpu):
        for r in range(R):
            ctx.set_positions(traj[r], replica=r)
        ctx.set_velocities_to_temperature(T, 99)
    for r in range(R):
        assert np.allclose(gpu.get_velocities(r), cpu.get_velocities(r), atol=1e-5 if precision == FP32 else 1e-12)
    for chunk in (1, 4, 5):
        gpu.step(chunk, T, 0.002, 1.0, 2026)
        cpu.step(chunk, T, 0.002, 1.0, 2026)
    assert gpu.step_count == cpu.step_count == 10
    for r in range(R):
        dx = np.abs(gpu.get_positions(r) - cpu.get_positions(r)).max()
        dv = np.abs(gpu.get_velocities(r) - cpu.get_velocities(r)).max()
        assert dx <= tol and dv <= tol * 500, f"replica {r}: |dx|={dx:.3e} |dv|={dv:.3e}"


def test_virtual_sites_recomputed_like_oracle():
    oa, traj, _ = golden_system("2lyz")
    gpu = _gpu(oa, 1, FP64)
    cpu = awsem_oracle.make_context(oa.system)
    x = traj[2].copy()
    x[sorted(oa.system.virtual_sites)] += 0.05
    for ctx in (gpu, cpu):
        ctx.set_positions(x, replica=0)
        ctx.compute_virtual_sites()
    assert np.allclose(gpu.get_positions(0), cpu.get_positions(0), atol=1e-12)
    assert np.allclose(gpu.get_positions(0), traj[2], atol=2e-3)   # the PDB/DCD stores 3-decimal Angstrom sites


def test_bad_arguments_are_rejected_with_messages():
    api = _capi.load_hip_library()
    ctx = _capi.Context(api, 10, 1)
    with pytest.raises(_capi.AwsemError, match="bad arguments"):
        ctx.add_harmonic_bonds(20, [(0, 10)], [0.1], [1.0])      # atom index out of range
    with pytest.raises(_capi.AwsemError, match="group"):
        ctx.add_harmonic_bonds(32, [(0, 1)], [0.1], [1.0])
    with pytest.raises(_capi.AwsemError, match="not finalized"):
        ctx.energy()
    ctx.finalize()
    with pytest.raises(_capi.AwsemError, match="already finalized"):
        ctx.add_harmonic_bonds(20, [(0, 1)], [0.1], [1.0])


@pytest.mark.parametrize("builders", ["lean", "full"])
@pytest.mark.parametrize("precision", [FP32, FP64])
@pytest.mark.parametrize("protein", ["1brs", "2lyz"])
def test_pair_lists_equal_brute_force_bit_exactly(protein, precision, builders, monkeypatch):
    """The engine's list builders against the oracle's O(N^2) scan with the distance screen evaluated in the same precision and
    operation order: identical sorted pair sets, with and without a skin.  Both families: the ones the lean step / evaluation kernels
    carry (two-pass hydrogen-bond scan, excluded-volume scan with the lane's site in registers; awsem_neighbors uses them on a handle
    that runs the lean kernels) and the diagonal enumeration of the full kernel (AWSEM_STEP_LEAN=0).  The reference keeps OpenMM's
    neighbour lists private, so there is no fixture: parity unpinned."""
    if builders == "full":
        monkeypatch.setenv("AWSEM_STEP_LEAN", "0")
    oa, traj, _ = golden_system(protein)
    gpu = _gpu(oa, 2, precision)
    cpu = _capi.Context(awsem_oracle.load(), oa.natoms, 2, precision=precision)
    oa.system.upload(cpu)
    for ctx in (gpu, cpu):
        ctx.set_positions(traj[0], replica=0)
        ctx.set_positions(traj[7], replica=1)
    n_terms = {_capi.LIST_EXCL: 1, _capi.LIST_CONTACT: 1, _capi.LIST_BETA: 3, _capi.LIST_PAP: 2}
    total = 0
    for kind, count in n_terms.items():
        for index in range(count):
            for replica in (0, 1):
                for skin in (0.0, 0.15):
                    a = gpu.neighbors(kind, index, replica, skin)
                    b = cpu.neighbors(kind, index, replica, skin)
                    assert a.shape == b.shape and np.array_equal(a, b), f"kind {kind} term {index} replica {replica} skin {skin}"
                    total += len(a)
    assert total > 1000


def test_trajectory_does_not_depend_on_list_skin(monkeypatch):
    """The evaluators re-apply the exact cutoffs and all sums are fixed point, so a trajectory must be bit-identical
    whatever the Verlet skin (= whenever the lists happen to be rebuilt): 0.02 nm rebuilds almost every step."""
    oa, traj, _ = golden_system("1ubq")
    out = []
    for skin in ("0.02", "0.15", "0.4"):
        monkeypatch.setenv("AWSEM_SKIN", skin)
        gpu = _gpu(oa, 3, FP32)
        gpu.set_positions(traj[3], replica=-1)
        T = np.array([300.0, 600.0, 800.0])
        gpu.set_velocities_to_temperature(T, 5)
        for chunk in (120, 80):
            gpu.step(chunk, T, 0.002, 1.0, 77)
        out.append(np.stack([gpu.get_positions(r) for r in range(3)]))
        gpu.close()
    assert np.isfinite(out[0]).all()
    assert np.abs(out[0][2] - traj[3]).max() > 0.05          # the chain really moved further than the small skins
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])


# --------------------------------------------------------------------------- minimiser (awsem_minimize)
# Parity unpinned by the reference: OpenMM's LocalEnergyMinimizer (L-BFGS) is only specified by its contract, a
# nearby local minimum whose RMS force is below the tolerance.  The device minimiser (FIRE) is checked against that
# contract, against the fp64 CPU restatement of the same algorithm, and for bit reproducibility.
@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol_rms", [(FP32, 10.0), (FP64, 10.0)])
def test_minimizer_reaches_the_force_tolerance(precision, tol_rms):
    oa, traj, _ = golden_system("1ubq")
    frames = [0, 5, 11]
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=len(frames), precision=precision)
    oa.system.upload(gpu)
    for r, f in enumerate(frames):
        gpu.set_positions(traj[f], replica=r)
    gpu.compute_virtual_sites()
    e0 = gpu.energy()
    it, rms = gpu.minimize(tol_rms, 20000)
    e1 = gpu.energy()
    massive = oa.system.masses > 0
    f = gpu.forces(redistribute=True)
    rms_check = np.sqrt((f[:, massive] ** 2).sum(axis=(1, 2)) / (3 * massive.sum()))
    assert np.all(it < 20000), it
    assert np.all(rms <= tol_rms) and np.all(rms_check <= tol_rms * 1.01), (rms, rms_check)
    assert np.all(e1 < e0 - 50.0), (e0, e1)            # the DCD frames are thermal structures: hundreds of kJ/mol above a minimum
    x = np.stack([gpu.get_positions(r) for r in range(len(frames))])
    assert np.isfinite(x).all()
    # virtual sites sit where their parents put them
    y = x.copy()
    gpu.compute_virtual_sites()
    assert np.allclose(y, np.stack([gpu.get_positions(r) for r in range(len(frames))]), atol=1e-5 if precision == FP32 else 1e-12)
    # bit reproducible
    for r, fr in enumerate(frames):
        gpu.set_positions(traj[fr], replica=r)
    gpu.compute_virtual_sites()
    it2, rms2 = gpu.minimize(tol_rms, 20000)
    assert np.array_equal(it, it2) and np.array_equal(rms, rms2)
    assert np.array_equal(x, np.stack([gpu.get_positions(r) for r in range(len(frames))]))


@pytest.mark.gpu
def test_minimizer_follows_the_oracle_restatement():
    """fp64 device FIRE vs the fp64 C restatement: same algorithm, so the first iterations agree closely and both end
    in minima of comparable depth (the paths separate slowly: summation order only)."""
    oa, traj, _ = golden_system("1ubq")
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=1, precision=FP64)
    oa.system.upload(gpu)
    cpu = awsem_oracle.make_context(oa.system, n_replicas=1)
    for c in (gpu, cpu):
        c.set_positions(traj[3], replica=0)
        c.compute_virtual_sites()
    for c in (gpu, cpu):
        c.minimize(1e-6, 25)       # 25 iterations, tolerance out of reach
    assert np.abs(gpu.get_positions(0) - cpu.get_positions(0)).max() < 1e-7
    ig, rg = gpu.minimize(10.0, 20000)
    ic, rc = cpu.minimize(10.0, 20000)
    assert rg[0] <= 10.0 and rc[0] <= 10.0
    eg, ec = gpu.energy()[0], cpu.energy()[0]
    assert abs(eg - ec) < 0.02 * abs(ec), (eg, ec, ig, ic)


@pytest.mark.parametrize("env", [{"AWSEM_FUSE_VS": "0"}, {"AWSEM_NO_REORDER": "1"}, {"AWSEM_FM_BUNDLES": "0"}, {"AWSEM_BLOCK": "256"}])
def test_alternative_kernel_paths_give_the_same_trajectory(env, monkeypatch):
    """The step kernel's optional paths -- virtual sites in a phase of their own, the caller's atom order kept, fragment
    memory bond by bond, four waves instead of sixteen -- change how the work is laid out, not what is summed: in fp64
    mode 60 steps end within rounding of the default path (fixed-point sums, same random streams)."""
    oa, traj, _ = golden_system("1ubq")
    T = np.array([300.0, 600.0])
    out = []
    for variant in (None, env):
        for k in ("AWSEM_FUSE_VS", "AWSEM_NO_REORDER", "AWSEM_FM_BUNDLES", "AWSEM_BLOCK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in (variant or {}).items():
            monkeypatch.setenv(k, v)
        gpu = _gpu(oa, 2, FP64)
        gpu.set_positions(traj[3])
        gpu.set_velocities_to_temperature(T, 5)
        gpu.step(60, T, 0.002, 1.0, 6)
        out.append((gpu.get_state()[0], gpu.energy_by_group(), gpu.forces()))
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-9
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-6 and np.abs(out[0][2] - out[1][2]).max() < 1e-5


@pytest.mark.gpu
def test_pacing_and_placement_leave_the_trajectories_alone(monkeypatch):
    """fp32 step kernel: which workgroup takes which replica (dealt anew every few launches by list length) and the priority bias a
    replica carries from launch to launch (csrc/awsem_kernels.h, pace_bias / awsem_place_kernel) decide WHEN a replica's instructions
    issue, never what they compute: 300 replicas (two rows of workgroups, the second one partial) over six launches end bit for bit
    where the plain rotation (AWSEM_PRIO=1) and the hardware's own order (AWSEM_PRIO=0) end."""
    oa, traj, _ = golden_system("1ubq")
    R = 300
    T = np.linspace(250.0, 800.0, R)
    out = []
    for prio, every in (("1", "0"), ("3", "2"), ("3", "1"), ("0", "0")):
        monkeypatch.setenv("AWSEM_PRIO", prio)
        monkeypatch.setenv("AWSEM_PLACE_EVERY", every)
        gpu = _gpu(oa, R, FP32)
        for r in range(R):
            gpu.set_positions(traj[r % len(traj)], replica=r)
        gpu.set_velocities_to_temperature(T, 5)
        for _ in range(6):
            gpu.step(20, T, 0.002, 1.0, 6)
        out.append(gpu.get_state())
        gpu.close()
    monkeypatch.delenv("AWSEM_PRIO", raising=False)
    monkeypatch.delenv("AWSEM_PLACE_EVERY", raising=False)
    for x, v in out[1:]:
        assert np.array_equal(x, out[0][0]) and np.array_equal(v, out[0][1])
    assert np.isfinite(out[0][0]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("workload, flavour", [("build_1r69", "lean instantiation"), ("build_t0958", "lean + helical instantiation"),
                                               ("build_2xov", "lean + membrane instantiation")])
def test_lean_step_kernel_follows_the_full_one(workload, flavour, monkeypatch, capfd):
    """The step kernel exists four times (csrc/awsem_kernels.h, evaluate_terms<LEAN>): the full one, one without the code of the rarer term
    shapes for systems that hold none of them (the default forces_setup.py list), that one plus a helical term (the many-memory
    configurations), and one with the z-dependent contact term and a membrane term instead (the membrane-protein set-up).  The same source expressions, but not the same machine code: the compiler contracts a * b + c into one rounding or
    two depending on what surrounds it (hipcc's default -ffp-contract=fast; with =on the three agree bit for bit and run 4 % slower), so the
    instantiations differ where fp32 rounds -- a force contribution now and then lands on the neighbouring 2^-12 of the fixed-point
    accumulators.  Checked: the library says which one it chose; after ONE step from the same state velocities agree to 1e-5 nm/ps and
    positions to 1e-6 nm, after 160 steps (list rebuilds, 250 ... 800 K) positions to 1e-3 nm (measured: 1e-6 ... 1.5e-4); each instantiation reproduces itself bit for bit."""
    from openawsem_amd import workloads
    oa = getattr(workloads, workload)()
    R = 12
    T = np.linspace(250.0, 800.0, R)
    monkeypatch.delenv("AWSEM_FORCE_CELL", raising=False)   # about the replica-resident path's kernel
    monkeypatch.setenv("AWSEM_INFO", "1")
    out = []
    for lean in ("1", "0", "1"):
        monkeypatch.setenv("AWSEM_STEP_LEAN", lean)
        capfd.readouterr()
        gpu = _gpu(oa, R, FP32)
        gpu.set_positions(oa.pdb.positions, replica=-1)
        gpu.set_velocities_to_temperature(T, 5)
        gpu.step(1, T, 0.002, 1.0, 6)
        first = gpu.get_state()
        for _ in range(4):
            gpu.step(40, T, 0.002, 1.0, 6)
        out.append(first + gpu.get_state())
        gpu.close()
        said = capfd.readouterr().err
        assert ("step kernel: " + (flavour if lean == "1" else "full instantiation")) in said, said
    lean, full, again = out
    assert all(np.array_equal(p, q) for p, q in zip(lean, again))
    assert np.abs(lean[0] - full[0]).max() < 1e-6 and np.abs(lean[1] - full[1]).max() < 1e-5
    assert np.abs(lean[2] - full[2]).max() < 1e-3
    assert np.isfinite(lean[2]).all() and not np.array_equal(lean[2][0], lean[2][R - 1])


@pytest.mark.gpu
def test_the_skin_rule_cannot_show_in_a_trajectory(monkeypatch):
    """The skin only decides how often the lists are rebuilt and how long they are (supersets of the interacting pairs either way):
    the rule that follows the lists' lengths (default), the rule in T alone (AWSEM_SKIN_ADAPT=0) and two pinned skins end twelve
    launches of a cooling ensemble bit for bit alike."""
    monkeypatch.delenv("AWSEM_FORCE_CELL", raising=False)
    oa, traj, _ = golden_system("1ubq")
    R = 40
    out = []
    for env in ({}, {"AWSEM_SKIN_ADAPT": "0"}, {"AWSEM_SKIN": "0.08"}, {"AWSEM_SKIN": "0.3"}):
        for k in ("AWSEM_SKIN_ADAPT", "AWSEM_SKIN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gpu = _gpu(oa, R, FP32)
        for r in range(R):
            gpu.set_positions(traj[r % len(traj)], replica=r)
        gpu.set_velocities_to_temperature(np.full(R, 700.0), 5)
        for it in range(12):
            gpu.step(25, np.full(R, 700.0 - 30.0 * it), 0.002, 1.0, 6)
        out.append(gpu.get_state())
        gpu.close()
    for k in ("AWSEM_SKIN_ADAPT", "AWSEM_SKIN"):
        monkeypatch.delenv(k, raising=False)
    for x, v in out[1:]:
        assert np.array_equal(x, out[0][0]) and np.array_equal(v, out[0][1])


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [FP32, FP64])
def test_lists_carried_from_launch_to_launch_change_nothing(precision, monkeypatch):
    """A step launch goes on with the pair lists, list lengths and reference positions its predecessor left in device memory
    (csrc/awsem_kernels.h, `reuse`) instead of rebuilding at its first step.  Lists are supersets of the interacting pairs and the sums
    are fixed point, so WHEN a list is rebuilt cannot show: launches interleaved with everything that must invalidate the carried lists
    (new positions for one replica, a minimisation, a warmer launch = larger skin) or must not (energies, forces, neighbour queries,
    new velocities) end bit for bit where AWSEM_LIST_REUSE=0 ends."""
    monkeypatch.delenv("AWSEM_FORCE_CELL", raising=False)
    oa, traj, _ = golden_system("1ubq")
    R = 6
    out = []
    for reuse in ("1", "0"):
        monkeypatch.setenv("AWSEM_LIST_REUSE", reuse)
        gpu = _gpu(oa, R, precision)
        for r in range(R):
            gpu.set_positions(traj[(3 * r) % len(traj)], replica=r)
        T = np.linspace(300.0, 700.0, R)
        gpu.set_velocities_to_temperature(T, 5)
        log = []
        for it in range(4):
            gpu.step(15, T - 20.0 * it, 0.002, 1.0, 6)          # cooling: the skin shrinks, the lists stay
        log.append(gpu.energy_by_group())                        # queries between launches: lists of their own
        gpu.forces()
        gpu.neighbors(_capi.LIST_CONTACT, 0, 1, 0.1)
        gpu.step(15, T - 80.0, 0.002, 1.0, 6)
        gpu.set_positions(traj[1], replica=2)                    # one replica moved from outside
        gpu.step(15, T - 80.0, 0.002, 1.0, 6)
        gpu.set_velocities_to_temperature(T + 100.0, 7)
        gpu.step(15, T + 100.0, 0.002, 1.0, 6)                   # warmer: larger skin
        gpu.step(15, T + 100.0, 0.002, 1.0, 6)
        log.append(gpu.energy_by_group())
        out.append((gpu.get_state(), log))
        gpu.close()
    monkeypatch.delenv("AWSEM_LIST_REUSE", raising=False)
    assert np.array_equal(out[0][0][0], out[1][0][0]) and np.array_equal(out[0][0][1], out[1][0][1])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)
    assert np.isfinite(out[0][0][0]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("knob", ["AWSEM_BETA_SCREEN", "AWSEM_PAP_SCREEN"])
def test_beta_screen_drops_only_exact_zeros(knob, monkeypatch):
    """fp32: the beta terms skip a pair once its Gaussian factor is below e^-X, X chosen at finalize so that every force component of a
    skipped pair is under 2^-16 kJ/mol/nm, an eighth of what the Q19.12 accumulators round to zero (csrc/awsem_hip.hip, beta block;
    X is about 18 with the reference's tables).  Round 1's X = 30 must therefore give the same forces and the same trajectory BIT FOR
    BIT; the energies, which sum in Q31.32, may differ by the dropped tails (< 1e-6 kJ/mol here).  P-AP likewise: a pair is skipped
    once any of its three switches has an argument below A = log(2^-16 / (|c|max eta2)), about -16.6, where round 1 used -30.
    (The argument is about the replica-resident path's Q19.12 accumulators: the test pins that path.)"""
    monkeypatch.delenv("AWSEM_FORCE_CELL", raising=False)
    oa, traj, _ = golden_system("1ubq")
    T = np.array([300.0, 700.0])
    out = []
    for screen in (None, "30"):
        monkeypatch.delenv(knob, raising=False)
        if screen:
            monkeypatch.setenv(knob, screen)
        gpu = _gpu(oa, 2, FP32)
        gpu.set_positions(traj[0], replica=0)
        gpu.set_positions(traj[9], replica=1)
        f0 = gpu.forces(redistribute=False)
        e0 = gpu.energy_by_group()
        gpu.set_velocities_to_temperature(T, 5)
        gpu.step(300, T, 0.002, 1.0, 6)
        out.append((f0, e0, gpu.get_state()))
        gpu.close()
    monkeypatch.delenv(knob, raising=False)
    assert np.array_equal(out[0][0], out[1][0])                       # forces
    assert np.array_equal(out[0][2][0], out[1][2][0]) and np.array_equal(out[0][2][1], out[1][2][1])   # positions, velocities after 300 steps
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-6 and np.abs(out[0][1][:, 27 if "BETA" in knob else 28]).min() > 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["1ubq", "2xov"])
def test_cropped_fragment_memory_rows_drop_only_exact_zeros(workload, monkeypatch):
    """fp32: the force walk of the fragment-memory term reads rows cropped to the bins whose slope can show in a Q19.12 accumulator
    (csrc/awsem_hip.hip, fragment-memory block: k |slope| >= 2^-15, a quarter of what rounds to zero; the full row is the reference's,
    templateTerms.py:273-281).  With AWSEM_FM_CROP=0 the same walk reads full rows: forces and a 300-step trajectory must agree BIT FOR
    BIT, and so must the energies, which are always evaluated from the full rows.  2xov is the configuration the crop is for (6 908
    bonds, a 27.6 MB table against 4 MB of L2).  (Q19.12 is the replica-resident path's: the test pins that path.)"""
    monkeypatch.delenv("AWSEM_FORCE_CELL", raising=False)
    if workload == "1ubq":
        oa, traj, _ = golden_system("1ubq")
        starts = [traj[0], traj[9]]
    else:
        from openawsem_amd import workloads
        oa = workloads.build_2xov()
        x0 = np.asarray(oa.pdb.positions, dtype=np.float64)
        starts = [x0, x0 + np.random.default_rng(3).normal(0, 0.005, x0.shape)]
    T = np.array([300.0, 700.0])
    out = []
    for crop in (None, "0"):
        monkeypatch.delenv("AWSEM_FM_CROP", raising=False)
        if crop:
            monkeypatch.setenv("AWSEM_FM_CROP", crop)
        gpu = _gpu(oa, 2, FP32)
        for r in range(2):
            gpu.set_positions(starts[r], replica=r)
        f0 = gpu.forces(redistribute=False)
        e0 = gpu.energy_by_group()
        gpu.set_velocities_to_temperature(T, 5)
        gpu.step(300, T, 0.002, 1.0, 6)
        out.append((f0, e0, gpu.get_state()))
        gpu.close()
    monkeypatch.delenv("AWSEM_FM_CROP", raising=False)
    assert np.array_equal(out[0][0], out[1][0])                       # forces
    assert np.array_equal(out[0][1], out[1][1]) and np.abs(out[0][1][:, 23]).min() > 1.0   # energies (group 23 = Fragment)
    assert np.array_equal(out[0][2][0], out[1][2][0]) and np.array_equal(out[0][2][1], out[1][2][1])   # positions, velocities after 300 steps


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [FP32, FP64])
@pytest.mark.parametrize("skin", ["0.03", "0.25"])
def test_queries_after_a_launch_walk_the_launch_s_own_lists(precision, skin, monkeypatch):
    """Energies and forces asked for right after a step launch (a reporter; the energies of a replica-exchange sweep, scripts/mm_remd.py:
    133-149) walk the pair lists that launch left in device memory instead of building their own (awsem_eval_kernel, `reuse`).  With a
    skin of 0.03 nm nearly every replica ends a launch further than skin/2 from where its lists were built: the query then rebuilds them
    in place and hands them to the next launch; with 0.25 nm none does.  Either way energies (also of a subset of the groups), forces
    and the trajectory that follows must equal AWSEM_QUERY_REUSE=0 bit for bit."""
    monkeypatch.delenv("AWSEM_FORCE_CELL", raising=False)
    monkeypatch.setenv("AWSEM_SKIN", skin)
    oa, traj, _ = golden_system("1ubq")
    R = 5
    out = []
    for reuse in ("1", "0"):
        monkeypatch.setenv("AWSEM_QUERY_REUSE", reuse)
        gpu = _gpu(oa, R, precision)
        for r in range(R):
            gpu.set_positions(traj[(2 * r) % len(traj)], replica=r)
        T = np.linspace(300.0, 800.0, R)
        gpu.set_velocities_to_temperature(T, 5)
        log = []
        for it in range(3):
            gpu.step(20, T, 0.002, 1.0, 6)
            log.append(gpu.energy((1 << 22) | (1 << 27)))         # contact and beta only: a stale replica still rebuilds every list
            log.append(gpu.forces())
            gpu.step(20, T, 0.002, 1.0, 6)
            log.append(gpu.energy_by_group())
        out.append((gpu.get_state(), log))
        gpu.close()
    monkeypatch.delenv("AWSEM_QUERY_REUSE", raising=False)
    assert np.array_equal(out[0][0][0], out[1][0][0]) and np.array_equal(out[0][0][1], out[1][0][1])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)
    assert np.isfinite(out[0][0][0]).all() and np.abs(out[0][1][0]).min() > 1.0
