"""Does pipelining a 32-scene step as k sub-batches of 32 / k scenes shorten the fill and drain of a short timed region?
usage: python tools/probe/split_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
for k in (1, 2, 4):
    parts = [pts[i * (32 // k):(i + 1) * (32 // k)].contiguous() for i in range(k)]
    det = SADDetector(cfg, w, dev, overlap_fps=True, n_fps_streams=3, n_main_streams=2)
    det.autotune(parts[0])
    def run(steps, depth=6 * k):
        evs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for p in parts:
                out, ev = det.submit(p)
                evs.append(ev)
                if len(evs) > depth:
                    evs.pop(0).synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    run(5)
    t20 = min(run(20) for _ in range(3))
    t200 = run(200)
    print(f"{k} sub-batches of {32 // k}: 20 steps {t20:.3f} ms/step = {32e3 / t20:.0f} scenes/s; 200 steps {t200:.3f} ms/step = {32e3 / t200:.0f} scenes/s", flush=True)
