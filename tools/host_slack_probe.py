"""Is the pipelined step bound by the enqueueing thread?  Dose-response, no profiler: the bench's submit loop (two main streams, sampling streams,
rotating resident batches) with a busy-wait of d microseconds added to the host side of every step.  Host-bound: ms/step grows by d from the first
increment.  GPU-bound: flat until d reaches the host's slack, then slope 1 - the knee is the slack.
usage: python tools/host_slack_probe.py [kitti|nuscenes] [f32|bf16] [fps streams] [steps per point]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sad_amd
from sad_amd import config, synth
from sad_amd.detector import SADDetector
name = sys.argv[1] if len(sys.argv) > 1 else "kitti"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
nf = int(sys.argv[3]) if len(sys.argv) > 3 else (6 if dtype == "bf16" else 3)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
B, NB = 32, (4 if name == "kitti" else 2)
dev = torch.device("cuda:0")
cfg = config.KITTI if name == "kitti" else config.NUSCENES
w = synth.make_weights(cfg, 0)
mk = synth.make_batch if name == "kitti" else synth.make_nuscenes_batch
batches = [torch.from_numpy(mk(k * B, B, cfg.n_points)).to(dev) for k in range(NB)]
det = SADDetector(cfg, w, dev, overlap_fps=True, n_fps_streams=nf, n_main_streams=2, dtype=dtype)
det.autotune(batches[0])
depth = max(6, nf + 2)
def run(delay_us, n):
    evs = []
    for k in range(8):                               # warm-up at this delay
        evs.append(det.submit(batches[k % NB])[1])
    torch.cuda.synchronize(); evs.clear()
    host = 0.0
    t0 = time.perf_counter()
    for k in range(n):
        h0 = time.perf_counter()
        out, ev = det.submit(batches[k % NB])
        h1 = time.perf_counter()
        host += h1 - h0
        if delay_us:
            t_end = h1 + delay_us * 1e-6
            while time.perf_counter() < t_end:
                pass
        evs.append(ev)
        if len(evs) > depth:
            evs.pop(0).synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, host / n * 1e3
delays = [0, 50, 100, 150, 200, 300, 400]
print(f"{name} {dtype} B={B}, {NB} rotating batches, {nf} sampling streams, queue depth {depth}, {steps} steps per point")
print("delay us | ms/step (ascending pass, descending pass) | host ms inside submit()")
up = [run(d, steps) for d in delays]
down = [run(d, steps) for d in reversed(delays)][::-1]
base = (up[0][0] + down[0][0]) / 2
for d, a, b in zip(delays, up, down):
    print(f"{d:8d} | {a[0]:.3f}  {b[0]:.3f}  (+{(a[0] + b[0]) / 2 - base:+.3f} over no delay) | {a[1]:.3f}  {b[1]:.3f}", flush=True)
