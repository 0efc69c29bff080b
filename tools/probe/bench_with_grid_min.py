"""bench.py with ops.GRID_MIN_POINTS set first (the ball query of scenes with fewer points takes the brute-force scan):
usage: python tools/probe/bench_with_grid_min.py POINTS -- <bench.py arguments>"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
thr = int(sys.argv[1])
args = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else sys.argv[2:]
import sad_amd
from sad_amd import ops
ops.GRID_MIN_POINTS = thr
sys.argv = [os.path.join(root, "bench.py")] + args
runpy.run_path(sys.argv[0], run_name="__main__")
