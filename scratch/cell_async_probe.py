"""FORCE_CELL awsem_run with 16 replicas, reports every 10 steps: does the graph capture survive the reporter thread?"""
import os, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openawsem_amd import data_path
from openawsem_amd.scripts import mm_run
tmp = tempfile.mkdtemp()
d = os.path.join(tmp, "1r69")
shutil.copytree(str(data_path.data / "1r69"), d, ignore=shutil.ignore_patterns("fraglib", "frags.mem"))
shutil.copy(os.path.join(d, "1r69.fasta"), os.path.join(d, "crystal_structure.fasta"))
seq = "".join(ln.strip() for ln in open(os.path.join(d, "1r69.fasta")) if not ln.startswith(">"))
open(os.path.join(d, "charge.txt"), "w").write("".join(f"{i} {1.0 if a in 'RK' else (-1.0 if a in 'DE' else 0.0)}\n" for i, a in enumerate(seq)))
os.chdir(d)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    mm_run.main(["1r69", "--to", os.path.join(tmp, "run0"), "-s", "4000", "--tempStart", "700", "--tempEnd", "300", "--replicas", "16", "--seed", "11",
                 "-f", os.path.join(os.path.dirname(mm_run.__file__), "forces_setup.py")])
print("ok", os.environ.get("AWSEM_ASYNC_REPORT"))
