"""awsem_energy_by_group_noalloc (include/awsem_hip.h): the allocation-free variant of awsem_energy_by_group.

It runs the same kernels on caller-owned device buffers with its pair lists in the library's static __device__ array, so
its energies must equal awsem_energy_by_group's BIT FOR BIT (both precisions: every sum is fixed point).  Two calls in a
row, the second with a smaller input whose output buffer is poisoned first; then a call with more replicas than the handle
has, and the size limit.  The device buffers and the stream come from the HIP runtime the library itself runs on
(libamdhip64.so.7, through ctypes -- the one instance in the process, see _capi._share_torchs_hip_runtime); torch is not needed
here.  Collected last (test_zz_*): the round-end run is `pytest -x`."""
import ctypes as C

import numpy as np
import pytest

from openawsem_amd import _capi
from tests.common import golden_system


class _Hip:
    """The handful of HIP runtime calls a caller of the entry point needs (allocation happens HERE, outside the entry point)."""

    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so.7")     # already loaded by libawsem_hip.so: the same instance
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.lib.hipStreamDestroy.argtypes = [C.c_void_p]
        self.lib.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
        self.lib.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        self.lib.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        self.lib.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.hipGraphExecDestroy.argtypes = [C.c_void_p]
        self.lib.hipGraphDestroy.argtypes = [C.c_void_p]
        self.bufs = []

    def ok(self, rc):
        assert rc == 0, f"HIP error {rc}"

    def to_device(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        p = C.c_void_p()
        self.ok(self.lib.hipMalloc(C.byref(p), a.nbytes))
        self.ok(self.lib.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1))      # hipMemcpyHostToDevice
        self.bufs.append(p)
        return p

    def to_host(self, p, shape):
        a = np.empty(shape, dtype=np.float64)
        self.ok(self.lib.hipMemcpy(a.ctypes.data_as(C.c_void_p), p, a.nbytes, 2))      # hipMemcpyDeviceToHost
        return a

    def stream(self):
        s = C.c_void_p()
        self.ok(self.lib.hipStreamCreate(C.byref(s)))
        return s

    def close(self, stream):
        self.ok(self.lib.hipStreamDestroy(stream))
        for p in self.bufs:
            self.ok(self.lib.hipFree(p))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [0, 1])
def test_noalloc_entry_point_equals_energy_by_group(precision, monkeypatch):
    monkeypatch.delenv("AWSEM_FORCE_CELL", raising=False)   # the entry point belongs to the replica-resident path (it says so on the other)
    oa, traj, _ = golden_system("1ubq")
    api = _capi.load_hip_library()
    n1, n2, n3 = 8, 3, 12
    gpu = _capi.Context(api, oa.natoms, n_replicas=n1, precision=precision)
    oa.system.upload(gpu)
    for r in range(n1):
        gpu.set_positions(traj[r], replica=r)
    want1 = gpu.energy_by_group()
    assert np.abs(want1).max() > 100.0
    hip = _Hip()
    stream = hip.stream()
    nan = lambda n: hip.to_device(np.full((n, 32), np.nan))   # noqa: E731

    x1, e1 = hip.to_device(traj[:n1]), nan(n1)
    gpu.energy_by_group_noalloc(n1, x1.value, e1.value, stream.value)
    hip.ok(hip.lib.hipStreamSynchronize(stream))
    assert np.array_equal(hip.to_host(e1, (n1, 32)), want1)

    # second call: fewer replicas, other frames (the static array still holds the first call's lists)
    frames = [11, 9, 4]
    for r, k in enumerate(frames):
        gpu.set_positions(traj[k], replica=r)
    want2 = gpu.energy_by_group()[:n2]
    x2, e2 = hip.to_device(traj[frames]), nan(n2)
    gpu.energy_by_group_noalloc(n2, x2.value, e2.value, stream.value)
    hip.ok(hip.lib.hipStreamSynchronize(stream))
    assert np.array_equal(hip.to_host(e2, (n2, 32)), want2)
    assert not np.array_equal(want2, want1[:n2])

    # more replicas than the handle has (n_replicas is per call), on the null stream
    x3, e3 = hip.to_device(traj[:n3]), nan(n3)
    gpu.energy_by_group_noalloc(n3, x3.value, e3.value, None)
    hip.ok(hip.lib.hipStreamSynchronize(None))
    got3 = hip.to_host(e3, (n3, 32))
    assert np.array_equal(got3[:n1], want1) and np.isfinite(got3).all()

    # under stream capture (what the entry point is for): captured once, replayed twice, the output poisoned in between
    x4, e4 = hip.to_device(traj[:n1]), nan(n1)
    graph, exe = C.c_void_p(), C.c_void_p()
    hip.ok(hip.lib.hipStreamBeginCapture(stream, 1))                     # hipStreamCaptureModeThreadLocal
    gpu.energy_by_group_noalloc(n1, x4.value, e4.value, stream.value)
    hip.ok(hip.lib.hipStreamEndCapture(stream, C.byref(graph)))
    hip.ok(hip.lib.hipGraphInstantiate(C.byref(exe), graph, None, None, 0))
    assert np.isnan(hip.to_host(e4, (n1, 32))).all()                     # capturing ran nothing
    for _ in range(2):
        hip.ok(hip.lib.hipGraphLaunch(exe, stream))
        hip.ok(hip.lib.hipStreamSynchronize(stream))
        assert np.array_equal(hip.to_host(e4, (n1, 32)), want1)
        poison = np.full((n1, 32), np.nan)
        hip.ok(hip.lib.hipMemcpy(e4, poison.ctypes.data_as(C.c_void_p), poison.nbytes, 1))
    hip.ok(hip.lib.hipGraphExecDestroy(exe))
    hip.ok(hip.lib.hipGraphDestroy(graph))

    # beyond the static array: a clear error, nothing launched
    with pytest.raises(_capi.AwsemError) as err:
        gpu.energy_by_group_noalloc(1 << 20, x1.value, e1.value, stream.value)
    assert err.value.code == -4 and "scratch" in str(err.value)
    with pytest.raises(_capi.AwsemError):
        gpu.energy_by_group_noalloc(0, x1.value, e1.value, stream.value)
    with pytest.raises(_capi.AwsemError):
        gpu.energy_by_group_noalloc(n1, 0, e1.value, stream.value)
    hip.close(stream)
    gpu.close()
