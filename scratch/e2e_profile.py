import cProfile, pstats, sys, os, runpy
sys.argv = ["e2e_run.py", sys.argv[1] if len(sys.argv) > 1 else "1024"]
here = os.path.dirname(os.path.abspath(__file__))
cProfile.run("runpy.run_path(os.path.join(here, 'e2e_run.py'), run_name='__main__')", "/tmp/e2e.prof")
pstats.Stats("/tmp/e2e.prof").sort_stats("cumulative").print_stats(35)
