import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openawsem_amd import _capi, workloads
oa = workloads.build_2xov()
ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, 256)
oa.system.upload(ctx)
ctx.set_positions(oa.pdb.positions)
for kind, name, n in ((_capi.LIST_EXCL, "excl", 1), (_capi.LIST_CONTACT, "contact", 1), (_capi.LIST_BETA, "beta", 3), (_capi.LIST_PAP, "pap", 2)):
    for idx in range(n):
        t = time.time(); p = ctx.neighbors(kind, idx, 0, 0.15); dt = time.time() - t
        t = time.time(); p = ctx.neighbors(kind, idx, 0, 0.15); dt = time.time() - t
        print(name, idx, "pairs", len(p), "wall ms %.3f" % (dt * 1e3))
