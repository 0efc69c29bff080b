"""Is a stage dispatch slower right behind the rest of the detector step than back to back?  Runs the SA3 stage
dispatch twice directly behind a serial detector pass (same stream) and times each with events."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth, _lib
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=False)
det.set_geometry(json.load(open("profiles/r02_geometry.json")))
tr = {}
det(pts, tr); torch.cuda.synchronize()
xyz, feat, new_xyz = tr["sa2"]["new_xyz"], tr["sa2"]["out"], tr["sa3"]["new_xyz"]
st = cfg.stages[2]
idxs, cnts = ops.ball_query_multi(st.radii, st.nsamples, xyz, new_xyz, return_counts=True)
wss = ops.rowscan_multi(idxs, cnts, xyz.shape[1])
out = torch.zeros(32, st.npoint, 768, device=dev)
calls = [(det.stages[2].branches[i], xyz, feat, new_xyz, idxs[i], out, 256 * i, cnts[i], wss[i]) for i in range(3)]
ev = lambda: torch.cuda.Event(enable_timing=True)
res = []
for rep in range(6):
    det(pts)
    e = [ev() for _ in range(4)]
    e[0].record(); ops.grouped_multi(calls); e[1].record(); ops.grouped_multi(calls); e[2].record(); ops.grouped_multi(calls); e[3].record()
    torch.cuda.synchronize()
    res.append([e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(3)])
print("SA3 stage dispatch behind a detector pass: 1st / 2nd / 3rd (us):", np.round(np.median(np.array(res), axis=0), 1))
# and with an idle gap (host sleep) in front
import time
res = []
for rep in range(6):
    torch.cuda.synchronize(); time.sleep(0.05)
    e = [ev() for _ in range(3)]
    e[0].record(); ops.grouped_multi(calls); e[1].record(); ops.grouped_multi(calls); e[2].record()
    torch.cuda.synchronize()
    res.append([e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(2)])
print("after 50 ms idle: 1st / 2nd (us):", np.round(np.median(np.array(res), axis=0), 1))
