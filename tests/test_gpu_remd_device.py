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


@pytest.mark.parametrize("cell", ["0", "1"])
@pytest.mark.parametrize("precision", [0, 1])
def test_device_sweeps_equal_host_sweeps(precision, cell, monkeypatch):
    monkeypatch.setenv("AWSEM_FORCE_CELL", cell)
    oa, traj, _ = golden_system("1ubq")
    n, iters, steps = 12, 12, 25
    T = np.linspace(300.0, 322.0, n)           # 2 K apart: plenty of accepted swaps for a 76-residue chain
    # host path: energies to the host, decisions in numpy, factors back to the device
    host = ensemble.ReplicaExchange(T, seed=5)
    a = _start(oa, traj, n, precision)
    a.set_velocities_to_temperature(host.local_temperatures(), 3)
    slots_host = []
    for it in range(iters):
        a.step(steps, host.local_temperatures(), 0.002, 1.0, 9)
        a.scale_velocities(host.exchange(a.energy()))
        slots_host.append(host.slot.copy())
    # device path: nothing leaves the device until the end
    dev = ensemble.ReplicaExchange(T, seed=5)
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
    gpu.close()


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
