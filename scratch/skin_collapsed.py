import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openawsem_amd import _capi, workloads
oa = workloads.build_1r69()
R = 1024
ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, R)
oa.system.upload(ctx)
ctx.set_positions(oa.pdb.positions)
T = np.full(R, 300.0)
ctx.set_velocities_to_temperature(T, 1)
ctx.step(20000, T, 0.002, 1.0, 1); ctx.synchronize()
x = ctx.get_positions(0); ca = x[oa.ca]
ms = []
for _ in range(10):
    ctx.step(50, T, 0.002, 1.0, 1); ms.append(ctx.last_kernel_ms())
print("skin", os.environ.get("AWSEM_SKIN"), "Rg %.2f nm" % np.sqrt(((ca - ca.mean(0)) ** 2).sum(1).mean()), "ms/launch %.4f" % np.median(ms), "-> %.2fM steps/s" % (R * 50 / np.median(ms) / 1e3))
