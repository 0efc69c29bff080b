"""Host cost of one SADDetector.submit with the GPU idle (nothing to wait behind), and where it goes (cProfile).
usage: python tools/host_profile.py [kitti|nuscenes] [f32|bf16] [batch] [fps streams]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sad_amd
from sad_amd import config, synth
from sad_amd.detector import SADDetector
name = sys.argv[1] if len(sys.argv) > 1 else "kitti"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
nf = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda:0")
cfg = config.KITTI if name == "kitti" else config.NUSCENES
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy((synth.make_batch if name == "kitti" else synth.make_nuscenes_batch)(0, B, cfg.n_points)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=True, n_fps_streams=nf, n_main_streams=2, dtype=dtype)
det.autotune(pts)
def idle_submit(label):
    for _ in range(det._plan_ring + 2):      # (with step plans: every ring slot recorded)
        det.submit(pts)
    torch.cuda.synchronize()
    ts = []
    for _ in range(48):
        t0 = time.perf_counter()
        det.submit(pts)
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    ts.sort()
    print(f"{name} {dtype} B={B} [{label}]: submit with an idle GPU: min {ts[0]*1e3:.3f} median {ts[len(ts)//2]*1e3:.3f} ms"
          f"  (plan replays so far {det.plan_replays}, refused: {det.plan_refused})", flush=True)
det.use_plans = False
idle_submit("eager")
det.use_plans = True
idle_submit("step plans")
if len(sys.argv) > 5 and sys.argv[5] == "noprofile":
    raise SystemExit(0)
pr = cProfile.Profile()
for _ in range(30):
    pr.enable()
    det.submit(pts)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
