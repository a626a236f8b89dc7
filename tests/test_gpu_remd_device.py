"""Replica exchange with the bookkeeping on the device (include/awsem_hip.h: awsem_remd_*) against the host-side sweep of
openawsem_amd/ensemble.py (itself checked against the closed-form Metropolis criterion in tests/test_ensemble.py).

The reference delegates REMD to openmmtools (scripts/mm_remd.py:133-149): PARITY UNPINNED.  Both paths draw the same
Philox numbers and apply the same velocity factors, so the two runs must stay bit-identical: same ladder assignment after
every sweep, same acceptance counts, same coordinates."""
import numpy as np
import pytest

from openawsem_amd import _capi, ensemble
from tests.common import golden_system

pytestmark = pytest.mark.gpu


def _start(oa, traj, n, precision):
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=n, precision=precision)
    oa.system.upload(gpu)
    for r in range(n):
        gpu.set_positions(traj[r % len(traj)], replica=r)
    return gpu


@pytest.mark.parametrize("mixing", ["neighbours", "swap-all"])
@pytest.mark.parametrize("cell", ["0", "1"])
@pytest.mark.parametrize("precision", [0, 1])
def test_device_sweeps_equal_host_sweeps(precision, cell, mixing, monkeypatch):
    monkeypatch.setenv("AWSEM_FORCE_CELL", cell)
    oa, traj, _ = golden_system("1ubq")
    n, iters, steps = 12, 12, 25
    T = np.linspace(300.0, 322.0, n)           # 2 K apart: plenty of accepted swaps for a 76-residue chain
    # host path: energies to the host, decisions in numpy, factors back to the device
    host = ensemble.ReplicaExchange(T, seed=5, mixing=mixing)
    a = _start(oa, traj, n, precision)
    a.set_velocities_to_temperature(host.local_temperatures(), 3)
    slots_host = []
    for it in range(iters):
        a.step(steps, host.local_temperatures(), 0.002, 1.0, 9)
        a.scale_velocities(host.exchange(a.energy()))
        slots_host.append(host.slot.copy())
    # device path: nothing leaves the device until the end
    dev = ensemble.ReplicaExchange(T, seed=5, mixing=mixing)
    b = _start(oa, traj, n, precision)
    b.set_velocities_to_temperature(dev.local_temperatures(), 3)
    dev.attach(b)
    slots_dev = []
    for it in range(iters):
        b.step_remd(steps, 0.002, 1.0, 9)
        dev.exchange_on_device()
        slots_dev.append(b.remd_state()[0].copy())       # (a synchronising query, for the comparison only)
    slot, att, acc, T_local = b.remd_state()
    assert host.accepted.sum() >= 5 and np.any(host.slot != np.arange(n))
    assert np.array_equal(np.array(slots_host), np.array(slots_dev))
    assert np.array_equal(att, host.attempted) and np.array_equal(acc, host.accepted)
    assert np.array_equal(T_local, T[host.slot])
    xa, va = a.get_state()
    xb, vb = b.get_state()
    assert np.array_equal(xa, xb) and np.array_equal(va, vb)
    assert np.array_equal(dev.pull().slot, host.slot) and dev.iteration == host.iteration
    a.close(); b.close()


def test_queries_between_sweeps_leave_the_ladder_temperatures_alone():
    """The device keeps the ladder's temperatures between sweeps; kinetic-energy and velocity-factor calls use a buffer of their own
    (they once shared it: a StateDataReporter between two sweeps would have turned kinetic energies into temperatures)."""
    oa, traj, _ = golden_system("1ubq")
    n = 6
    T = np.linspace(300.0, 350.0, n)
    rex = ensemble.ReplicaExchange(T, seed=5)
    gpu = _start(oa, traj, n, 0)
    gpu.set_velocities_to_temperature(rex.local_temperatures(), 3)
    rex.attach(gpu)
    gpu.step_remd(10, 0.002, 1.0, 9)
    before = gpu.remd_state()[3].copy()
    ke = gpu.kinetic_energy()
    gpu.scale_velocities(np.ones(n))
    assert np.array_equal(gpu.remd_state()[3], before) and np.array_equal(before, T) and np.all(ke > 0)
    gpu.step_remd(10, 0.002, 1.0, 9)
    assert np.isfinite(gpu.get_state()[0]).all()
    # neither do a Maxwell-Boltzmann draw nor an ordinary step at other temperatures (they once uploaded into the ladder's buffer: the
    # next step_remd would have run at them while slot[] said otherwise)
    other = np.full(n, 777.0)
    gpu.set_velocities_to_temperature(other, 4)
    assert np.array_equal(gpu.remd_state()[3], T)
    gpu.step(5, other, 0.002, 1.0, 9)
    gpu.synchronize()
    assert np.array_equal(gpu.remd_state()[3], T)
    # ... and the dynamics that follows is the ladder's: identical to a handle that never saw the foreign temperatures
    twin = _start(oa, traj, n, 0)
    twin.set_velocities_to_temperature(rex.local_temperatures(), 3)
    ensemble.ReplicaExchange(T, seed=5).attach(twin)
    x, v = gpu.get_state()
    twin.set_state(x, v)
    twin.step_count = gpu.step_count
    for c in (gpu, twin):
        c.step_remd(10, 0.002, 1.0, 9)
    assert np.array_equal(gpu.get_state()[0], twin.get_state()[0])
    # a second set-up replaces the first (new ladder, fresh counters) instead of piling up buffers
    T2 = np.linspace(310.0, 400.0, n)
    gpu.remd_setup(T2, local_start=0, gather_index=None, seed=6)
    slot, att, acc, T_local = gpu.remd_state()
    assert np.array_equal(T_local, T2) and np.array_equal(slot, np.arange(n)) and not att.any() and not acc.any()
    gpu.step_remd(5, 0.002, 1.0, 9)
    gpu.remd_energies(0xFFFFFFFF)
    gpu.remd_sweep(0)
    assert np.array_equal(np.sort(gpu.remd_state()[3]), T2)
    gpu.close(); twin.close()


@pytest.mark.parametrize("precision", [0, 1])
def test_the_rescale_of_a_sweep_is_the_same_whoever_takes_it_in(precision):
    """A sweep leaves its velocity factors on the device; the next step launch multiplies them in as it loads the velocities, and any
    other reader of the velocities (a state query, the kinetic energy, a snapshot) settles them with a launch first.  Both ways give the
    same bits, a query sees the rescaled velocities at once, and two sweeps in a row leave the product of their factors."""
    oa, traj, _ = golden_system("1ubq")
    n = 8
    T = np.linspace(300.0, 314.0, n)
    runs = []
    for query in (False, True):
        rex = ensemble.ReplicaExchange(T, seed=5)
        gpu = _start(oa, traj, n, precision)
        gpu.set_velocities_to_temperature(rex.local_temperatures(), 3)
        rex.attach(gpu)
        seen = []
        for it in range(6):
            gpu.step_remd(25, 0.002, 1.0, 9)
            before = gpu.get_state()[1]
            T_before = gpu.remd_state()[3].copy()
            rex.exchange_on_device()
            if query:
                ke = gpu.kinetic_energy()
                v = gpu.get_state()[1]
                T_after = gpu.remd_state()[3]
                f = np.sqrt(T_after / T_before)
                real = np.float32 if precision == 0 else np.float64
                want = (before.astype(real) * f.astype(real)[:, None, None]).astype(np.float64)
                assert np.array_equal(v, want)
                assert np.all(ke > 0)
                seen.append(np.any(f != 1.0))
        if query:
            assert any(seen)          # (some sweep did move a temperature)
        runs.append(gpu.get_state())
        if query:   # two sweeps with no launch between them: the factors multiply (one rounding instead of two)
            v0 = gpu.get_state()[1]
            T0 = gpu.remd_state()[3].copy()
            rex.exchange_on_device()
            rex.exchange_on_device()
            v2 = gpu.get_state()[1]
            f = np.sqrt(gpu.remd_state()[3] / T0)
            assert np.allclose(v2, v0 * f[:, None, None], rtol=3e-7 if precision == 0 else 1e-15, atol=0)
            # ... and nothing is applied twice: a further query returns the same velocities, a launch starts from them
            assert np.array_equal(gpu.get_state()[1], v2)
        gpu.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])


def test_sharded_ladder_with_a_gather_index_takes_the_same_decisions():
    """Two handles play two ranks inside one process: each holds a block of the ladder (4 + 3 replicas) and writes its energies into
    its padded block of one shared device buffer, as two ranks would through the all-gather (shards padded to 4: replica r of rank 1
    sits at 4 + (r - 4)... with a hole at index 7 unused).  Both sweep kernels must take the decisions the host-side sweep takes on
    the same gathered energies, and hand their replicas the temperatures that follow."""
    import ctypes as C
    oa, traj, _ = golden_system("1ubq")
    n, iters, steps, pad = 7, 8, 20, 4
    T = np.linspace(300.0, 315.0, n)
    hip = C.CDLL("libamdhip64.so.7")            # the runtime the library is linked against
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), 8 * 2 * pad) == 0
    index = np.array([0, 1, 2, 3, 4, 5, 6], dtype=np.int32)      # rank 0: r; rank 1: pad + (r - 4)
    parts = []
    for lo, hi in ((0, 4), (4, 7)):
        c = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=hi - lo, precision=1)
        oa.system.upload(c)
        for r in range(lo, hi):
            c.set_positions(traj[r], replica=r - lo)
        c.set_velocities_to_temperature(T[lo:hi], 3 + lo)
        c.remd_setup(T, local_start=lo, gather_index=index, seed=8)
        parts.append((c, lo, hi))
    host = ensemble.ReplicaExchange(T, seed=8)
    gathered = np.zeros(2 * pad)
    for it in range(iters):
        for c, lo, hi in parts:
            c.step_remd(steps, 0.002, 1.0, 9)
            c.remd_energies(0xFFFFFFFF, buf.value + 8 * lo)
        for c, lo, hi in parts:
            c.synchronize()
        assert hip.hipMemcpy(gathered.ctypes.data_as(C.c_void_p), buf, gathered.nbytes, 2) == 0
        U = gathered[index]
        assert np.allclose(U[:4], parts[0][0].energy()) and np.allclose(U[4:], parts[1][0].energy())
        host.exchange(U)
        for c, lo, hi in parts:
            c.remd_sweep(it, buf.value)
        for c, lo, hi in parts:
            slot, att, acc, T_local = c.remd_state()
            assert np.array_equal(slot, host.slot) and np.array_equal(acc, host.accepted)
            assert np.array_equal(T_local, T[host.slot[lo:hi]])
    assert host.accepted.sum() >= 3
    assert hip.hipFree(buf) == 0
    for c, _, _ in parts:
        c.close()


def test_energies_handed_over_in_torch_memory_give_the_same_sweeps():
    """The hand-off of ensemble.exchange_on_device with several ranks (ensemble.py: ctx.remd_energies(mask, g.local.data_ptr()) ...
    ctx.remd_sweep(it, g.all.data_ptr())): the engine writes its energies into memory that TORCH allocated and the sweep kernel reads
    the gathered buffer from there.  torch ships its own copy of the HIP runtime next to the one libawsem_hip.so links against, so
    this crosses runtimes inside one process; the engine's stream is joined (ctx.synchronize()) before torch touches the buffer and
    torch's stream before the sweep reads it.  Must equal the NULL-buffer path (the engine's own buffer) bit for bit: ladder
    assignment, counters, temperatures, coordinates."""
    import torch
    oa, traj, _ = golden_system("1ubq")
    n, iters, steps = 8, 10, 20
    T = np.linspace(300.0, 316.0, n)
    runs = []
    for torch_buffers in (False, True):
        ctx = _start(oa, traj, n, 0)
        ctx.set_velocities_to_temperature(T, 3)
        # a one-rank _Gather has the layout of the multi-rank one (padded block + length word, replica r at index[r])
        g = ensemble._Gather(n, 1, 0, torch.device("cuda", 0)) if torch_buffers else None
        ctx.remd_setup(T, local_start=0, gather_index=None if g is None else g.index, seed=4)
        slots = []
        for it in range(iters):
            ctx.step_remd(steps, 0.002, 1.0, 9)
            if g is None:
                ctx.remd_energies(0xFFFFFFFF)
                ctx.remd_sweep(it)
            else:
                ctx.remd_energies(0xFFFFFFFF, g.local.data_ptr())
                ctx.synchronize()
                g.all.copy_(g.local)                      # what the all-gather does for a world of one, on torch's stream
                torch.cuda.current_stream().synchronize()
                assert np.array_equal(g.all.cpu().numpy()[g.index], ctx.energy())
                ctx.remd_sweep(it, g.all.data_ptr())
            slots.append(ctx.remd_state()[0].copy())
        runs.append((np.array(slots), ctx.remd_state(), ctx.get_state()))
        ctx.close()
    (sa, ra, xa), (sb, rb, xb) = runs
    assert np.array_equal(sa, sb) and np.any(sa[-1] != np.arange(n))
    for a, b in zip(ra, rb):
        assert np.array_equal(a, b)
    assert np.array_equal(xa[0], xb[0]) and np.array_equal(xa[1], xb[1])


@pytest.mark.parametrize("mixing", ["neighbours", "swap-all"])
def test_replica_exchange_sweeps_on_the_2xov_membrane_system(mixing):
    """BASELINE.json configs[3] itself: the 276-residue membrane protein (membrane_term + z-dependent contact, one replica per CU),
    a 16-rung ladder 300-330 K, exchanges after every 40 steps.  Device-side sweeps against the host-side ones on the same energies:
    identical ladder assignment after every sweep, identical counters, identical coordinates at the end."""
    from openawsem_amd import workloads
    oa = workloads.build_2xov()
    n, iters, steps = 16, 6, 40
    T = np.linspace(300.0, 330.0, n)
    runs = []
    for on_device in (False, True):
        rex = ensemble.ReplicaExchange(T, seed=12, mixing=mixing)
        ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=n, precision=0)
        oa.system.upload(ctx)
        ctx.set_positions(oa.pdb.positions)
        ctx.minimize(10.0, 0)
        ctx.set_velocities_to_temperature(T, 3)
        slots = []
        if on_device:
            rex.attach(ctx)
        for it in range(iters):
            if on_device:
                ctx.step_remd(steps, 0.002, 1.0, 9)
                rex.exchange_on_device()
                slots.append(ctx.remd_state()[0].copy())
            else:
                ctx.step(steps, rex.local_temperatures(), 0.002, 1.0, 9)
                ctx.scale_velocities(rex.exchange(ctx.energy()))
                slots.append(rex.slot.copy())
        if on_device:
            rex.pull()
        runs.append((np.array(slots), rex.attempted.copy(), rex.accepted.copy(), ctx.get_state()))
        ctx.close()
    (sa, ta, aa, xa), (sb, tb, ab, xb) = runs
    assert np.array_equal(sa, sb) and np.array_equal(ta, tb) and np.array_equal(aa, ab)
    assert aa.sum() > 0 and np.any(sa[-1] != np.arange(n))
    assert np.array_equal(xa[0], xb[0]) and np.array_equal(xa[1], xb[1]) and np.isfinite(xa[0]).all()
