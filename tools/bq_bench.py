"""Ball-query timing on the KITTI SA1 / SA2 shapes (B = 32): HIP events around sad_ball_query_grid_f32
(build + query), run under tools/trace_one.sh for the per-kernel split.  usage: python tools/bq_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth
dev = torch.device("cuda:0")
from sad_amd import _lib
if os.environ.get("BQ_VARIANT"): _lib.set_option("bq_variant", int(os.environ["BQ_VARIANT"]))
cfg = config.KITTI
dense = len(sys.argv) > 1 and sys.argv[1] == "dense"      # 20 m x 20 m scenes: most centroids have > 128 candidates
nus = len(sys.argv) > 1 and sys.argv[1] == "nus"          # nuScenes-shaped: 65 536 points on 102 m x 102 m, 16 384 / 4 096 centroids
if nus:
    cfg = config.NUSCENES
    pts = torch.from_numpy(synth.make_nuscenes_batch(0, 32)).to(dev)
else:
    pts = torch.from_numpy((synth.make_dense_batch if dense else synth.make_batch)(0, 32)).to(dev)
xyz = pts[:, :, :3].contiguous()
M1, M2 = cfg.stages[0].npoint, cfg.stages[1].npoint
c1 = ops.gather_xyz(xyz, ops.fps(xyz, M1))
c2 = c1[:, :M2].contiguous()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st1, st2 = cfg.stages[0], cfg.stages[1]
t1 = timeit(lambda: ops.ball_query_multi(st1.radii, st1.nsamples, xyz, c1, return_counts=True))
t2 = timeit(lambda: ops.ball_query_multi(st2.radii, st2.nsamples, c1, c2, return_counts=True))
work = config.work_per_scene(cfg)
b1 = 32 * sum(cfg.n_points * 12 + M1 * 12 + M1 * s * 4 for s in st1.nsamples)
print("dense scenes" if dense else ("nuScenes-shaped scenes" if nus else "KITTI-shaped scenes"))
print(f"SA1 ball query ({cfg.n_points} -> {M1} x 3 radii, 32 scenes): {t1:.1f} us = {b1 / t1 / 1e3:.0f} GB/s of algorithmic bytes ({b1 / 1e6:.1f} MB)")
print(f"SA2 ball query ({M1} -> {M2} x 3 radii, 32 scenes): {t2:.1f} us")
