"""Ball query of the small stages (KITTI SA3: 1 024 points, 512 centroids, radii 1.6 / 3.2 / 4.8): brute-force scan against the grid kernels
(ops.GRID_MIN_POINTS decides).  Same indices and counts required; us per call, 32 scenes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sad_amd
from sad_amd import config, ops, synth
dev = torch.device("cuda:0")
cfg = config.KITTI
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
xyz = pts[:, :, :3].contiguous()
cs = []
cur = xyz
for st in cfg.stages:
    c = ops.gather_xyz(cur, ops.fps(cur, st.npoint)); cs.append((cur, c, st)); cur = c
def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, (src, cen, st) in zip(("sa1", "sa2", "sa3"), cs):
    N = src.shape[1]
    res = {}
    for thr in (1 << 30, 1):                  # brute force, grid
        ops.GRID_MIN_POINTS = thr
        idx, cnt = ops.ball_query_multi(st.radii, st.nsamples, src, cen, return_counts=True)
        t = min(timeit(lambda: ops.ball_query_multi(st.radii, st.nsamples, src, cen, return_counts=True)) for _ in range(3))
        res[thr] = (idx, cnt, t)
    same = all(torch.equal(a, b) for a, b in zip(res[1 << 30][0], res[1][0])) and all(torch.equal(a, b) for a, b in zip(res[1 << 30][1], res[1][1]))
    print(f"{name}: N {N} M {cen.shape[1]} radii {st.radii}: brute force {res[1 << 30][2]:.1f} us, grid {res[1][2]:.1f} us, identical: {same}", flush=True)
