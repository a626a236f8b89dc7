"""cProfile of scratch/e2e_run.py's awsem_run (second run in the process, so imports and first-touch costs are out)."""
import cProfile, pstats, os, shutil, sys, tempfile, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openawsem_amd import data_path
from openawsem_amd.scripts import mm_run
R = 1024
def once(profile):
    tmp = tempfile.mkdtemp()
    d = os.path.join(tmp, "1r69")
    shutil.copytree(str(data_path.data / "1r69"), d, ignore=shutil.ignore_patterns("fraglib", "frags.mem"))
    shutil.copy(os.path.join(d, "1r69.fasta"), os.path.join(d, "crystal_structure.fasta"))
    seq = "".join(ln.strip() for ln in open(os.path.join(d, "1r69.fasta")) if not ln.startswith(">"))
    open(os.path.join(d, "charge.txt"), "w").write("".join(f"{i} {1.0 if a in 'RK' else (-1.0 if a in 'DE' else 0.0)}\n" for i, a in enumerate(seq)))
    os.chdir(d)
    args = ["1r69", "--to", os.path.join(tmp, "run0"), "-s", "20000", "--replicas", str(R), "--seed", "7", "-f", os.path.join(os.path.dirname(mm_run.__file__), "forces_setup.py")]
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        if profile:
            pr = cProfile.Profile(); pr.enable(); mm_run.main(args); pr.disable()
        else:
            mm_run.main(args)
    dt = time.time() - t0
    print(f"{dt:.2f} s")
    if profile:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
    os.chdir("/"); shutil.rmtree(tmp)
once(False); once(True)
