"""End-to-end awsem_run of the 1r69 annealing ensemble (default 2e4 steps, 400 stages, all reporters): wall time."""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openawsem_amd import data_path
from openawsem_amd.scripts import mm_run
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tmp = tempfile.mkdtemp()
d = os.path.join(tmp, "1r69")
shutil.copytree(str(data_path.data / "1r69"), d, ignore=shutil.ignore_patterns("fraglib", "frags.mem"))
shutil.copy(os.path.join(d, "1r69.fasta"), os.path.join(d, "crystal_structure.fasta"))
seq = "".join(ln.strip() for ln in open(os.path.join(d, "1r69.fasta")) if not ln.startswith(">"))
open(os.path.join(d, "charge.txt"), "w").write("".join(f"{i} {1.0 if a in 'RK' else (-1.0 if a in 'DE' else 0.0)}\n" for i, a in enumerate(seq)))
os.chdir(d)
t0 = time.time()
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    sim = mm_run.main(["1r69", "--to", os.path.join(tmp, "run0"), "-s", "20000", "--replicas", str(R), "--seed", "7",
                       "-f", os.path.join(os.path.dirname(mm_run.__file__), "forces_setup.py")])
dt = time.time() - t0
loop = float(open(os.path.join(tmp, "run0", "time.dat")).read())
print(f"  of which the annealing loop (time.dat): {loop:.3f} s; set-up, minimisation, first frames, final files: {dt - loop:.3f} s")
print(f"awsem_run 1r69 x {R} replicas, 20000 steps, 400 stages: {dt:.2f} s wall -> {R * 20000 / dt / 1e6:.2f} M replica-steps/s end to end")
shutil.rmtree(tmp)
