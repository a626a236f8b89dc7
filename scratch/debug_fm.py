import sys; sys.path.insert(0,'.')
import numpy as np
from openawsem_amd import _capi
from oracle import awsem_oracle
from tests.common import *
oa, traj, _ = golden_system("1mbn")
gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, 1, 0, 0); oa.system.upload(gpu)
cpu = awsem_oracle.make_context(oa.system, 1)
x = traj[0]
gpu.set_positions(x, 0); cpu.set_positions(x, 0)
g = 1 << 23
fg, fc = gpu.forces(g, False)[0], cpu.forces(g, False)[0]
d = np.linalg.norm(fg - fc, axis=1)
print("norm fc", np.linalg.norm(fc), "norm diff", np.linalg.norm(fg-fc))
worst = np.argsort(d)[-6:]
print(worst, d[worst])
fm = [f for f in oa.system.forces if f.kind == "fragment_memory"][0]
for a in worst[-2:]:
    for b,(i,j) in enumerate(fm.pairs):
        if a in (i,j):
            r = np.linalg.norm(x[i]-x[j]); r32 = np.linalg.norm((x[i]-x[j]).astype(np.float32))
            idx = int(np.floor(r/0.01)); 
            if abs(r/0.01 - round(r/0.01)) < 1e-3: print("bond", b, i, j, "r", r, r32, "idx", idx, "slopes", (fm.table[b, idx]-fm.table[b, idx-1])/0.01, (fm.table[b, idx+1]-fm.table[b, idx])/0.01)
