"""Host time to ENQUEUE one step (SADDetector.submit, nothing waited for) next to the pipelined step time: is the pipeline
bound by the Python / launch path?   usage: python tools/host_enqueue_time.py [kitti|nuscenes] [f32|bf16] [batch] [fps streams]"""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth
from sad_amd.detector import SADDetector
name = sys.argv[1] if len(sys.argv) > 1 else "kitti"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
nf = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
cfg = config.KITTI if name == "kitti" else config.NUSCENES
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy((synth.make_batch if name == "kitti" else synth.make_nuscenes_batch)(0, B, cfg.n_points)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=True, n_fps_streams=nf, n_main_streams=2, dtype=dtype)
det.autotune(pts)
for _ in range(4):
    det.submit(pts)
torch.cuda.synchronize()
# (1) enqueue only: 6 steps deep, nothing waited for
t0 = time.perf_counter()
for _ in range(6):
    det.submit(pts)
t_host = (time.perf_counter() - t0) / 6 * 1e3
torch.cuda.synchronize()
# (2) pipelined
evs, steps = [], 60
t0 = time.perf_counter()
for _ in range(steps):
    out, ev = det.submit(pts)
    evs.append(ev)
    if len(evs) > max(6, nf + 2):
        evs.pop(0).synchronize()
torch.cuda.synchronize()
t_pipe = (time.perf_counter() - t0) / steps * 1e3
print(f"{name} {dtype} B={B} fps streams {nf}: host enqueue {t_host:.3f} ms/step, pipelined step {t_pipe:.3f} ms ({B / t_pipe * 1e3:.0f} scenes/s)")
