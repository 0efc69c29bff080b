"""Debug: per-pass per-launch event intervals in the serial pass vs the one-main-stream overlapped pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import sad_amd
from sad_amd import config, ops, synth
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
cfg = config.KITTI
det = SADDetector(cfg, synth.make_weights(cfg, 0), dev)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det.autotune(pts)
for mode in ("serial", "iso", "serial_side_stream"):
    det.overlap_fps = mode == "iso"
    st = torch.cuda.Stream(device=dev) if mode == "serial_side_stream" else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        det(pts); torch.cuda.synchronize()
        for p in range(4):
            ops.LAUNCH_LOG = []
            t0 = time.perf_counter()
            det(pts, input_ready=(mode == "iso"))
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
            print(mode, p, f"host enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms",
                  " ".join(f"{n}={e0.elapsed_time(e1):.3f}" for k, n, e0, e1 in log if k == "mlp"))
