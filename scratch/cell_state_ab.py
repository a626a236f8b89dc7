"""Final state of a short cell-list run under two builds of the library: must agree bit for bit (argv: library A, library B)."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 2:   # child: run and print the digest
    sys.path.insert(0, ROOT)
    import numpy as np
    from openawsem_amd import _capi, workloads
    oa = workloads.build_synthetic(n_chains=16, seed=3, hbonds=True)
    ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=2)
    oa.system.upload(ctx)
    ctx.set_positions(oa.pdb.positions, replica=-1)
    ctx.set_velocities_to_temperature(np.array([300.0, 500.0]), 7)
    for it in range(6):
        ctx.step(37, np.array([300.0, 500.0]), 0.002, 1.0, 7)
    x, v = ctx.get_state()
    print(hashlib.sha256(x.tobytes() + v.tobytes()).hexdigest(), float(np.abs(x).max()))
else:
    out = []
    for lib in sys.argv[1:3]:
        env = dict(os.environ, AWSEM_HIP_LIBRARY=lib if lib != "-" else "")
        out.append(subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=300).stdout.strip())
        print(lib, out[-1])
    sys.exit(0 if out[0] == out[1] and out[0] else 1)
