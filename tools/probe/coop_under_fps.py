"""Measurement build only (build/libsad_cstamps.so): per-workgroup start / end of the cooperative SA3 dispatch while an
FPS kernel (32 workgroups of 1024 threads) runs on another stream."""
import os, sys, ctypes, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=False)
g = json.load(open(os.path.join(root, "profiles", "r02_geometry.json")))
det.set_geometry(g)
tr = {}
det(pts, tr); torch.cuda.synchronize()
xyz0 = pts[:, :, :3].contiguous()
xyz, feat, new_xyz = tr["sa2"]["new_xyz"], tr["sa2"]["out"], tr["sa3"]["new_xyz"]
st = cfg.stages[2]
idxs, cnts = ops.ball_query_multi(st.radii, st.nsamples, xyz, new_xyz, return_counts=True)
wss = ops.rowscan_multi(idxs, cnts, xyz.shape[1])
out = torch.zeros(32, st.npoint, 768, device=dev)
calls = [(det.stages[2].branches[i], xyz, feat, new_xyz, idxs[i], out, 256 * i, cnts[i], wss[i]) for i in range(3)]
L = ctypes.CDLL(os.environ["SAD_AMD_LIB"])
def report(tag):
    buf2 = (ctypes.c_ulonglong * (2048 * 4))(); assert L.sad_debug_read_coop_all(buf2) == 0
    a = np.array(buf2, dtype=np.uint64).reshape(2048, 4).astype(np.int64); a = a[a[:, 1] > 0]
    t0 = a[:, 0].min(); st_, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    life = en - st_
    hw = a[:, 3] & 0xFFFFFFFF; xcc = a[:, 3] >> 32
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"{tag}: workgroups {len(a)} on {len(set(key.tolist()))} CUs; start us p50 {np.median(st_):.1f} p90 {np.percentile(st_, 90):.1f} max {st_.max():.1f}; "
          f"end us p50 {np.median(en):.0f} p90 {np.percentile(en, 90):.0f} max {en.max():.0f}; lifetime us p10 {np.percentile(life, 10):.0f} p50 {np.median(life):.0f} "
          f"p90 {np.percentile(life, 90):.0f} p99 {np.percentile(life, 99):.0f} max {life.max():.0f}")
    late = st_ > 20
    print(f"    workgroups that started later than 20 us: {late.sum()} (their start p50 {np.median(st_[late]) if late.any() else 0:.0f} us)")
    slow = life > 1.25 * np.median(life)
    per_cu = np.bincount(np.unique(key, return_inverse=True)[1])
    print(f"    workgroups slower than 1.25 x median: {slow.sum()} on {len(set(key[slow].tolist()))} CUs; workgroups per CU: min {per_cu.min()} max {per_cu.max()}")
for _ in range(3):
    ops.grouped_multi(calls)
torch.cuda.synchronize()
report("alone")
side, main = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(side):
    ops.fps(xyz0, 4096); ops.fps(xyz0, 4096)
with torch.cuda.stream(main):
    for _ in range(3):
        ops.grouped_multi(calls)
torch.cuda.synchronize()
report("under one FPS kernel")
