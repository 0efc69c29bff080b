"""Measurement build only (build/libsad_cstamps.so): the cooperative chain kernel's per-workgroup start / end times and
clock INSIDE a serial detector pass (last geometry-4 launch of the step = the SA3 stage) and in a back-to-back loop."""
import os, sys, ctypes, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ["SAD_AMD_LIB"] = os.path.join(root, "build", "libsad_cstamps.so")
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth, _lib
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=False)
g = json.load(open(os.path.join(root, "profiles", "r02_geometry.json")))
g.update({"sa3.b0": 4, "sa3.b1": 4, "sa3.b2": 4, "cluster.b0": 3, "cluster.b1": 3})
det.set_geometry(g)
L = ctypes.CDLL(os.environ["SAD_AMD_LIB"])
def report(tag):
    buf = (ctypes.c_ulonglong * (64 * 16))(); assert L.sad_debug_read_coop_stamps(buf) == 0
    s = np.array(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64); s = s[s[:, 0] > 0]
    life = s[:, 9] - s[:, 8]; real = (s[:, 11] - s[:, 10]) / 100.0
    buf2 = (ctypes.c_ulonglong * (2048 * 4))(); assert L.sad_debug_read_coop_all(buf2) == 0
    a = np.array(buf2, dtype=np.uint64).reshape(2048, 4).astype(np.int64); a = a[a[:, 1] > 0]
    t0 = a[:, 0].min(); st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    tot = s[:, 4] - s[:, 0]
    print(f"{tag}: clock {life.sum() / real.sum():.0f} MHz; last tile {tot.mean():.0f} cycles; workgroups {len(a)}: start max {st.max():.1f} us; "
          f"end us p10 {np.percentile(en, 10):.0f} p50 {np.median(en):.0f} p90 {np.percentile(en, 90):.0f} max {en.max():.0f}")
for _ in range(3):
    det(pts); torch.cuda.synchronize()
report("in a serial detector pass")
tr = {}
det(pts, tr); torch.cuda.synchronize()
xyz, feat, new_xyz = tr["sa2"]["new_xyz"], tr["sa2"]["out"], tr["sa3"]["new_xyz"]
st = cfg.stages[2]
idxs, cnts = ops.ball_query_multi(st.radii, st.nsamples, xyz, new_xyz, return_counts=True)
wss = ops.rowscan_multi(idxs, cnts, xyz.shape[1])
out = torch.zeros(32, st.npoint, 768, device=dev)
calls = [(det.stages[2].branches[i], xyz, feat, new_xyz, idxs[i], out, 256 * i, cnts[i], wss[i]) for i in range(3)]
for _ in range(5):
    ops.grouped_multi(calls)
torch.cuda.synchronize()
report("back to back")
