import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openawsem_amd import _capi, workloads
name, R = sys.argv[1], int(sys.argv[2])
oa = {"2xov": workloads.build_2xov, "t0958": workloads.build_t0958, "1r69": workloads.build_1r69}[name]()
api = _capi.load_hip_library("scratch/libawsem_PROF.so")
ctx = _capi.Context(api, oa.natoms, R)
oa.system.upload(ctx)
ctx.set_positions(oa.pdb.positions)
T = np.full(R, 500.0)
ctx.set_velocities_to_temperature(T, 1)
for _ in range(4):
    ctx.step(50, T, 0.002, 1.0, 1)
    print("ms/launch", ctx.last_kernel_ms())
ctx.close()
