"""How long does the host take to ENQUEUE one step (no GPU wait)?  If this approaches the GPU time
per step the pipeline is launch-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sad_amd
from sad_amd import config, synth
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
cfg = config.KITTI
det = SADDetector(cfg, synth.make_weights(cfg, 0), dev)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det.autotune(pts)
for _ in range(5): det(pts, input_ready=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): det(pts, input_ready=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/20:.3f} ms/step ; total {1e3*(t2-t0)/20:.3f} ms/step")
