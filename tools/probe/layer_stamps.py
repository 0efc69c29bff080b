"""Measurement build only (libsad_lstamps.so, -DSAD_LAYER_STAMPS): per-wave ticks (s_memtime), real time
(s_memrealtime, 100 MHz) and k-loop ticks of the layer-streamed kernel's last three launches (= the three
layers of one chain).  usage: layer_stamps.py cluster.b1"""
import os, sys, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ["SAD_AMD_LIB"] = os.path.join(root, "build", "libsad_lstamps.so")
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
bi = int(name.split(".")[1][1:])
xyz, feat = tr["sa3"]["new_xyz"], tr["sa3"]["out"]
new_xyz = tr["cluster"]["cand"]
idxs, cnts = ops.ball_query_multi(cfg.cluster_scales, cfg.cluster_nsamples, xyz, new_xyz, tr["cluster"]["radius"], return_counts=True)
mlp = ops.PackedMLP(w[name], True, dev)
mlp.default_geometry = 3
out = torch.zeros(idxs[bi].shape[0], idxs[bi].shape[1], mlp.out_channels, device=dev)
for _ in range(4):
    mlp.grouped(xyz, feat, new_xyz, idxs[bi], out=out, cnt=cnts[bi])
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ["SAD_AMD_LIB"])
buf = (ctypes.c_ulonglong * (3 * 64 * 8))()
assert L.sad_debug_read_layer_stamps(buf) == 0
s = np.array(buf, dtype=np.uint64).reshape(3, 64, 8).astype(np.int64)
for l in range(3):
    r = s[l]
    ok = r[:, 2] > r[:, 0]
    ticks = (r[ok, 2] - r[ok, 0]); real = (r[ok, 3] - r[ok, 1])
    clk = ticks.sum() / real.sum() * 100.0
    print(f"launch slot {l}: waves {ok.sum()}, ticks per wave {ticks.mean():.0f}, real {real.mean() / 100:.1f} us -> clock {clk:.0f} MHz; "
          f"items per wave {r[ok, 4].mean():.1f}, k-loop ticks {r[ok, 5].sum() / max(1, ticks.sum()):.3f} of the wave's ticks, "
          f"k-loop ticks per item {r[ok, 5].sum() / max(1, r[ok, 4].sum()):.0f}")
buf2 = (ctypes.c_ulonglong * (4096 * 4))()
assert L.sad_debug_read_layer_all(buf2) == 0
a = np.array(buf2, dtype=np.uint64).reshape(4096, 4).astype(np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en, it = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2]
print(f"last launch: {len(a)} waves; start us: min {st.min():.1f} p50 {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f}; "
      f"end us: min {en.min():.1f} p10 {np.percentile(en, 10):.1f} p50 {np.median(en):.1f} p90 {np.percentile(en, 90):.1f} max {en.max():.1f}")
for k in sorted(set(it)):
    m = it == k
    print(f"   waves with {k} items: {m.sum()}, lifetime us mean {(en[m] - st[m]).mean():.1f}, end mean {en[m].mean():.1f} max {en[m].max():.1f}")
hw = a[:, 3] & 0xFFFFFFFF; xcc = a[:, 3] >> 32
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
key = (xcc * 8 + se) * 2 + sh
key = key * 16 + cu
print("distinct (xcc,se,sh,cu):", len(set(key.tolist())), " waves per CU: min", min(np.bincount(np.unique(key, return_inverse=True)[1])), "max", max(np.bincount(np.unique(key, return_inverse=True)[1])))
ks = key * 4 + simd
cnt_simd = np.bincount(np.unique(ks, return_inverse=True)[1])
print("waves per SIMD histogram:", np.bincount(cnt_simd))
print("mean end time by XCC:", [round(float(en[xcc == x].mean()), 1) for x in range(8)])
print("mean end time by SE :", [round(float(en[se == x].mean()), 1) for x in sorted(set(se.tolist()))])
print("mean end time by CU index:", [round(float(en[cu == x].mean()), 1) for x in sorted(set(cu.tolist()))])
wid = np.arange(len(en))
print("mean end time by wave id / 512:", [round(float(en[(wid // 512) == x].mean()), 1) for x in range(6)])
blk = wid // 4
print("mean end by (block % 8):", [round(float(en[(blk % 8) == x].mean()), 1) for x in range(8)])
print("mean end by wave-in-block:", [round(float(en[(wid % 4) == x].mean()), 1) for x in range(4)])
