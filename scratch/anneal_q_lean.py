"""tools/anneal_q.py with the dynamics on the LEAN step kernel: the Q term lives in a second handle that only evaluates."""
import contextlib, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openawsem_amd import _capi, workloads
from openawsem_amd.functionTerms import biasTerms
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps_per_stage = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
with contextlib.redirect_stdout(io.StringIO()):
    oa = workloads.build_1r69()
    oq = workloads.build_1r69()
    oq.addForcesWithDefaultForceGroup([biasTerms.q_value(oq, os.path.join(ROOT, "tests", "golden", "1r69.pdb"), forceGroup=1)])
api = _capi.load_hip_library()
ctx = _capi.Context(api, oa.natoms, n_replicas=R); oa.system.upload(ctx)
cq = _capi.Context(api, oq.natoms, n_replicas=R); oq.system.upload(cq)
ctx.set_positions(oa.pdb.positions); ctx.minimize(10.0, 0)
ctx.set_velocities_to_temperature(np.full(R, 800.0), 7)
t0 = time.time()
for i in range(400):
    T = 800.0 + (200.0 - 800.0) / 400 * i
    ctx.step(steps_per_stage, np.full(R, T), 0.002, 1.0, 7)
    if i % 100 == 99:
        x, _ = ctx.get_state(velocities=False); cq.set_state(x, None)
        q = cq.energy(1 << 1); e = ctx.energy()
        print(f"stage {i + 1:3d} T {T:5.0f} K  <Q> {q.mean():.3f}  max Q {q.max():.3f}  <E> {e.mean():9.1f} kJ/mol  ({time.time() - t0:.1f} s)", flush=True)
print("final Q quantiles (5, 25, 50, 75, 95 %):", np.round(np.quantile(q, [0.05, 0.25, 0.5, 0.75, 0.95]), 3), " fraction with Q > 0.5:", float((q > 0.5).mean()))
print(f"{R} replicas x {400 * steps_per_stage} steps in {time.time() - t0:.1f} s")
