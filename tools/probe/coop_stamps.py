"""Measurement build only (build/libsad_cstamps.so: bash tools/probe/build_stamps.sh cstamps mlp_coop.hip -DSAD_COOP_STAMPS):
s_memtime at the phase boundaries of the LAST tile of the first 64 waves of the cooperative chain kernel.
usage: coop_stamps.py sa3.b2"""
import os, sys, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ["SAD_AMD_LIB"] = os.path.join(root, "build", "libsad_cstamps.so")
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
xyz = tr[f"sa{si}"]["new_xyz"]; feat = tr[f"sa{si}"]["out"]
new_xyz = tr[stage]["new_xyz"]; st = cfg.stages[si]
idxs, cnts = ops.ball_query_multi(st.radii, st.nsamples, xyz, new_xyz, return_counts=True)
mlp = ops.PackedMLP(w[name], True, dev)
mlp.default_geometry = 4
out = torch.zeros(idxs[bi].shape[0], idxs[bi].shape[1], mlp.out_channels, device=dev)
for _ in range(3):
    mlp.grouped(xyz, feat, new_xyz, idxs[bi], out=out, cnt=cnts[bi])
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ["SAD_AMD_LIB"])
buf = (ctypes.c_ulonglong * (64 * 16))()
assert L.sad_debug_read_coop_stamps(buf) == 0
s = np.array(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
s = s[s[:, 0] > 0]
tot = s[:, 4] - s[:, 0]
print(f"waves {len(s)}: tile total {tot.mean():.0f} cycles (min {tot.min()}, max {tot.max()})")
print(f"  rows + layers 0/1 (196 fragments): {(s[:, 1] - s[:, 0]).mean():.0f}")
print(f"  in2 conversion + pool masks:       {(s[:, 2] - s[:, 1]).mean():.0f}")
print(f"  layer 2 k-loops (sum over tiles):  {s[:, 5].mean():.0f}")
print(f"  pooling + staging (sum):           {s[:, 6].mean():.0f}")
print(f"  last stage + flush:                {(s[:, 4] - s[:, 3]).mean():.0f}")
life = s[:, 9] - s[:, 8]; real = (s[:, 11] - s[:, 10]) / 100.0
print(f"wave lifetime: {life.mean():.0f} cycles = {real.mean():.1f} us (clock {life.sum() / real.sum():.0f} MHz), items per workgroup {s[:, 12].mean():.2f}; "
      f"cycles per item {(life / np.maximum(1, s[:, 12])).mean():.0f}")
buf2 = (ctypes.c_ulonglong * (2048 * 4))()
assert L.sad_debug_read_coop_all(buf2) == 0
a = np.array(buf2, dtype=np.uint64).reshape(2048, 4).astype(np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
print(f"workgroups {len(a)}: start us p10 {np.percentile(st, 10):.1f} p50 {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f}; "
      f"end us p10 {np.percentile(en, 10):.1f} p50 {np.median(en):.1f} p90 {np.percentile(en, 90):.1f} max {en.max():.1f}")
hw = a[:, 3] & 0xFFFFFFFF; xcc = a[:, 3] >> 32
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
early = st < 5.0
cnt = np.bincount(np.unique(key[early], return_inverse=True)[1])
print(f"distinct CUs {len(set(key.tolist()))}; workgroups that started in the first 5 us: {early.sum()} on {len(cnt)} CUs (per CU min {cnt.min()} max {cnt.max()})")
