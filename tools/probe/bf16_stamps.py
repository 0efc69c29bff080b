"""Measurement build only (bash tools/probe/build_stamps.sh brst mlp_bf16_reg.hip -DSAD_BR_STAMPS -> build/libsad_brst.so):
per-phase s_memtime sums over the tiles of the first 1024 waves of the register-resident bf16 chain kernel.
usage: bf16_stamps.py sa3 | cluster | sa2 | sa1   (the merged dispatch of the stage) or sa3.b2 (one chain)"""
import os, sys, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ["SAD_AMD_LIB"] = os.path.join(root, "build", "libsad_brst.so")
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth, _lib
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
name = sys.argv[1]
stage = name.split(".")[0]
only = int(name.split(".")[1][1:]) if "." in name else None
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=False, dtype="bf16")
tr = {}
det(pts, tr); torch.cuda.synchronize()
if stage == "cluster":
    xyz, feat = tr["sa3"]["new_xyz"], tr["sa3"]["out"]; new_xyz = tr["cluster"]["cand"]
    idxs, cnts = ops.ball_query_multi(cfg.cluster_scales, cfg.cluster_nsamples, xyz, new_xyz, tr["cluster"]["radius"], return_counts=True)
    mlps = cfg.cluster_mlps
else:
    si = int(stage[2]) - 1
    xyz = pts[:, :, :3].contiguous() if si == 0 else tr[f"sa{si}"]["new_xyz"]
    feat = pts[:, :, 3:] if si == 0 else tr[f"sa{si}"]["out"]
    new_xyz = tr[stage]["new_xyz"]; st = cfg.stages[si]
    idxs, cnts = ops.ball_query_multi(st.radii, st.nsamples, xyz, new_xyz, return_counts=True)
    mlps = st.mlps
nets = [ops.PackedMLPBf16(w[f"{stage}.b{i}"], True, dev) for i in range(len(mlps))]
for n in nets: n.default_geometry = 2
wss = ops.rowscan_multi(idxs, cnts, xyz.shape[1])
out = torch.zeros(idxs[0].shape[0], idxs[0].shape[1], sum(m[-1] for m in mlps), device=dev)
calls, off = [], 0
for n, idx, cnt, ws, m in zip(nets, idxs, cnts, wss, mlps):
    calls.append((n, xyz, feat, new_xyz, idx, out, off, cnt, ws)); off += m[-1]
if only is not None:
    calls = [calls[only]]
def run():
    if len(calls) > 1: ops.grouped_multi(calls)
    else:
        c = calls[0]; c[0].grouped(*c[1:5], out=c[5], col_off=c[6], cnt=c[7], ws=c[8])
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
L = ctypes.CDLL(os.environ["SAD_AMD_LIB"])
buf = (ctypes.c_ulonglong * (1024 * 16))()
assert L.sad_debug_read_br_stamps(buf) == 0
s = np.array(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
s = s[s[:, 3] > 0]
tiles = s[:, 3].sum()
print(f"{name}: launch {e0.elapsed_time(e1)*1e3:.0f} us; waves sampled {len(s)}, tiles per wave {s[:, 3].mean():.2f}, fragments per tile {s[:, 4].sum() / tiles:.0f}")
print(f"  per tile (cycles): rows+gather+layer0 {s[:, 0].sum() / tiles:.0f}, layer1 {s[:, 1].sum() / tiles:.0f}, layer2+pool {s[:, 2].sum() / tiles:.0f}; "
      f"MFMA floor {32 * s[:, 4].sum() / tiles:.0f}")
print(f"  layer 2: k-loops {s[:, 6].sum() / tiles:.0f}, pooling + output {s[:, 7].sum() / tiles:.0f}")
life, real = s[:, 12], s[:, 13] / 100.0
print(f"  wave lifetime {life.mean():.0f} cycles = {real.mean():.1f} us (max {real.max():.1f}; clock {life.sum() / real.sum():.0f} MHz); in tiles {s[:, :3].sum() / life.sum() * 100:.0f} %")
