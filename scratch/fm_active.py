"""How many fragment-memory bonds contribute a force the fp32 accumulators do not round to zero?  (hot extended state, compact state)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openawsem_amd import _capi, workloads
R = 64
oa = workloads.build_1r69()
fm = [f for f in oa.system.forces if type(f).__name__ == "FragmentMemoryForce"][0]
pairs = np.asarray(fm.pairs).reshape(-1, 2); table = np.asarray(fm.table); dr = fm.dr; k = fm.k_fm
slope = np.abs(k * np.diff(table, axis=1) / dr)            # [bond][bin]
ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, R)
oa.system.upload(ctx)
ctx.set_positions(oa.pdb.positions)
ctx.set_velocities_to_temperature(np.full(R, 800.0), 2026)
def report(tag, skin):
    x = ctx.get_state()[0]
    d = np.linalg.norm(x[:, pairs[:, 0]] - x[:, pairs[:, 1]], axis=2)      # [R][bond]
    nb = slope.shape[1]
    act_now, act_skin = [], []
    for r in range(0, R, 8):
        b = np.minimum((d[r] / dr).astype(int), nb - 1)
        act_now.append((slope[np.arange(len(b)), b] >= 2.0 ** -16).mean())
        lo = np.minimum(((d[r] - skin).clip(0) / dr).astype(int), nb - 1); hi = np.minimum(((d[r] + skin) / dr).astype(int), nb - 1)
        act_skin.append(np.mean([(slope[i, lo[i]:hi[i] + 1] >= 2.0 ** -16).any() for i in range(len(b))]))
    print(f"{tag}: active now {np.mean(act_now):.2f}, active within +-{skin} nm {np.mean(act_skin):.2f} of {len(pairs)} bonds")
report("start (extended)", 0.25)
for i in range(20):
    ctx.step(50, np.full(R, 800.0 - 1.5 * i), 0.002, 1.0, 7)
report("800 K after 1000 steps", 0.25)
for i in range(20, 150):
    ctx.step(50, np.full(R, 800.0 - 1.5 * i), 0.002, 1.0, 7)
report("575 K after 7500 steps", 0.2)
for i in range(400):
    ctx.step(250, np.full(R, 800.0 - 500.0 * i / 400), 0.002, 1.0, 7)
report("300 K compact", 0.10)
