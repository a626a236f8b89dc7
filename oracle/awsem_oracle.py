"""TEST INFRASTRUCTURE -- loader for the fp64 CPU oracle (oracle/awsem_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The oracle exports
the include/awsem_hip.h ABI under the prefix ``awsem_oracle_``, so the same term descriptors
(openawsem_amd.forces) can be uploaded to it and to the HIP library and the two compared.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE = os.path.join(_HERE, "awsem_oracle.c")
LIBRARY = os.path.join(_HERE, "libawsem_oracle.so")

_EXTRA = {
    "set_threads": (C.c_int, ["H", C.c_int]),
}


def build(force=False):
    """gcc -O2 -fopenmp -shared; rebuilt when the source is newer than the library."""
    if not force and os.path.exists(LIBRARY) and (not os.path.exists(SOURCE) or
                                                   os.path.getmtime(LIBRARY) >= os.path.getmtime(SOURCE)):
        return LIBRARY
    cmd = ["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=gnu11", "-Wall", "-o", LIBRARY, SOURCE, "-lm"]
    subprocess.run(cmd, check=True)
    return LIBRARY


_api = None


def load():
    global _api
    if _api is None:
        from openawsem_amd._capi import CApi
        if not os.path.exists(LIBRARY):
            build()
        _api = CApi(C.CDLL(LIBRARY), "awsem_oracle_", extra=_EXTRA)
        lib = _api.lib
        lib.awsem_oracle_gauss3.restype = None
        lib.awsem_oracle_gauss3.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_double)]
        lib.awsem_oracle_hb_fn_table.restype = C.c_int
        lib.awsem_oracle_hb_fn_table.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return _api


def make_context(system, n_replicas=1, threads=1):
    """Upload an openawsem_amd.forces.System to a fresh oracle context."""
    from openawsem_amd._capi import Context
    ctx = Context(load(), system.getNumParticles(), n_replicas)
    system.upload(ctx)
    if threads != 1:
        ctx._call("set_threads", int(threads))
    return ctx


def hb_fn_table(fn):
    """The dense table an openawsem_amd.forces.HbTableFn stands for, evaluated entry by entry by the oracle's restatement of the
    reference's formulas."""
    import numpy as np
    api = load()
    n = fn.nres
    out = np.empty((n, n))
    info = np.ascontiguousarray(fn.res_info, dtype=np.int32)
    coef = np.ascontiguousarray(fn.coef, dtype=np.float64)
    rc = api.lib.awsem_oracle_hb_fn_table(fn.kind, n, info.ctypes.data_as(C.POINTER(C.c_int32)), coef.ctypes.data_as(C.POINTER(C.c_double)),
                                          out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out
