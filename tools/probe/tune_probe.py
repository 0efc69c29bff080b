import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["SAD_TUNE_DEBUG"] = "1"
import sad_amd, torch
from sad_amd import config, synth
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
cfg = config.KITTI
det = SADDetector(cfg, synth.make_weights(cfg, 0), dev)
pts = torch.from_numpy(synth.make_batch(0, 32, cfg.n_points)).to(dev)
g = det.autotune(pts)
print({k: g[k] for k in ("cluster.b0", "cluster.b1")})
