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
