import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openawsem_amd import _capi, workloads
oa = workloads.build_1r69()
ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, 1024)
oa.system.upload(ctx)
ctx.set_positions(oa.pdb.positions)
e0 = ctx.energy()[0]
t = time.time(); it, rms = ctx.minimize(10.0, 0); dt = time.time() - t
print("1r69 x1024: iterations", it[:4], "rms", rms[:4], "E", e0, "->", ctx.energy()[0], "kernel ms", ctx.last_kernel_ms(), "wall", dt)
