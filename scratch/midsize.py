"""Throughput of the golden proteins (76-218 residues, all 16 terms) at various workgroup sizes and replica counts."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openawsem_amd import _capi
from tests.common import golden_system

for protein in sys.argv[1].split(","):
    oa, traj, _ = golden_system(protein)
    for R in (256, 512, 1024):
        for block in ("", "256", "512", "1024"):
            if block:
                os.environ["AWSEM_BLOCK"] = block
            else:
                os.environ.pop("AWSEM_BLOCK", None)
            ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=R)
            oa.system.upload(ctx)
            ctx.set_positions(traj[0])
            T = np.full(R, 500.0)
            ctx.set_velocities_to_temperature(T, 1)
            ctx.step(50, T, 0.002, 1.0, 1)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                ctx.step(50, T, 0.002, 1.0, 1)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            print(f"{protein} nres {oa.nres} natoms {oa.natoms} R {R} block {block or 'auto':>4s}: {R * 500 / dt / 1e6:7.2f} M replica-steps/s", flush=True)
            ctx.close()
