"""Measurement build only (libsad_stamps.so, -DSAD_REG_STAMPS): s_memtime at the phase boundaries of the
register-resident chain kernel; prints the cycles each phase of a tile takes.  usage: reg_stamps.py sa3.b2"""
import os, sys, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ["SAD_AMD_LIB"] = os.path.join(root, "build", "libsad_stamps.so")
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth, _lib
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
for kv in filter(None, os.environ.get("SAD_OPTS", "").split(",")):
    k, v = kv.split("="); _lib.set_option(k, int(v))
name = sys.argv[1]
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=False)
tr = {}
det(pts, tr); torch.cuda.synchronize()
stage, br = name.split("."); bi = int(br[1:]); si = int(stage[2]) - 1
xyz = pts[:, :, :3].contiguous() if si == 0 else tr[f"sa{si}"]["new_xyz"]
feat = pts[:, :, 3:] if si == 0 else tr[f"sa{si}"]["out"]
new_xyz = tr[stage]["new_xyz"]; st = cfg.stages[si]
idxs, cnts = ops.ball_query_multi(st.radii, st.nsamples, xyz, new_xyz, return_counts=True)
mlp = ops.PackedMLP(w[name], True, dev)
out = torch.zeros(idxs[bi].shape[0], idxs[bi].shape[1], mlp.out_channels, device=dev)
_lib.set_option("mlp_force", 2)
for _ in range(3):
    mlp.grouped(xyz, feat, new_xyz, idxs[bi], out=out, cnt=cnts[bi])
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ["SAD_AMD_LIB"])
buf = (ctypes.c_ulonglong * (64 * 64))()
assert L.sad_debug_read_stamps(buf) == 0
s = np.array(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
names = {0: "start", 1: "row map + gather + LDS image + ring preload", 10: "layer 0/1 done + in2 conversion", 11: "pool masks", 30: "end"}
for wv in (0, 1, 5, 17):
    r = s[wv]
    if r[0] == 0: continue
    print(f"wave slot {wv}: tile total {r[30] - r[0]} cycles")
    print(f"   start -> gathered/ring: {r[1] - r[0]}")
    for o in range(4):
        print(f"   L0 tile {o}: wait+k-loop {r[3 + 2*o] - r[2 + 2*o]}   (prev feed: {r[2+2*o] - (r[3+2*o-2] if o else r[1])})")
    print(f"   last feed + convert: {r[10] - r[9]}   masks: {r[11] - r[10]}")
    for o in range(8):
        prev = r[11] if o == 0 else r[13 + 2 * (o - 1)]
        print(f"   L2 tile {o}: k-loop {r[12 + 2*o] - prev}   pool+store {r[13 + 2*o] - r[12 + 2*o]}")
print("grid (workgroups):", s[0][62])
for wv in (0, 1, 5, 17, 33):
    r = s[wv]
    print(f"wave slot {wv}: kernel start {r[63] - s[:, 63][s[:, 63] > 0].min()}; tiles (start,end rel. to wave start):",
          [(int(r[40 + 2 * k] - r[63]), int(r[41 + 2 * k] - r[63])) for k in range(11) if r[40 + 2 * k] > 0])
st = s[:, 63]
print("block start times (cycles, grouped by clock domain = XCD; block = slot // 4):")
for x in range(8):
    blks = [b for b in range(16) if b % 8 == x]
    base = min(st[b * 4] for b in blks)
    print(f"  XCD {x}:", [(b, int(st[b * 4] - base), int(s[b * 4][41 + 4] - st[b*4]) if s[b*4][45] > 0 else None) for b in blks])
