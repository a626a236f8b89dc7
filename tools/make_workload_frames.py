#!/usr/bin/env python3
"""Fixed coordinate sets for tests/test_gpu_workloads.py: for each BASELINE.json workload the start structure plus two
snapshots (after 400 and 3000 steps) of a Langevin run at 500 K of the fp64 CPU ORACLE -- so the GPU parity test evaluates
committed coordinates and does not depend on the GPU integrator at all.  Run once here (CPU only, ~1 min):

    python tools/make_workload_frames.py        -> tests/golden/workload_frames_<name>.npy  ([3][natoms][3] float64, nm)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from openawsem_amd import workloads  # noqa: E402
from oracle import awsem_oracle  # noqa: E402

BUILDERS = {
    "1r69": workloads.build_1r69,
    "1r69-fraglib": lambda: workloads.build_1r69(fragment_library=True),
    "T0958": workloads.build_t0958,
    "2xov": workloads.build_2xov,
}


def main():
    awsem_oracle.build()
    for name, build in BUILDERS.items():
        oa = build()
        cpu = awsem_oracle.make_context(oa.system, n_replicas=1)
        cpu.set_positions(oa.pdb.positions)
        cpu.set_velocities_to_temperature([500.0], 3)
        frames, done = [], 0
        for n in (0, 400, 3000):
            if n > done:
                cpu.step(n - done, [500.0], 0.002, 1.0, 3)
                done = n
            frames.append(cpu.get_positions(0))
        frames = np.array(frames)
        assert np.isfinite(frames).all()
        out = os.path.join(ROOT, "tests", "golden", f"workload_frames_{name}.npy")
        np.save(out, frames)
        print(name, frames.shape, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
