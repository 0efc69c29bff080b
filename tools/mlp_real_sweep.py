"""Times geometry codes for one grouped chain of the KITTI topology on the REAL ball-query output
(B = 32).  usage: python tools/mlp_real_sweep.py sa3.b2 25831 15831 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth, _lib
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
for kv in filter(None, os.environ.get("SAD_OPTS", "").split(",")):      # e.g. SAD_OPTS=mlp_dyn_slots=1
    k, v = kv.split("=")
    _lib.set_option(k, int(v))
name = sys.argv[1]
codes = [int(c) for c in sys.argv[2:]] or [0]
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=False)
tr = {}
det(pts, tr); torch.cuda.synchronize()
stage, br = name.split(".")
bi = int(br[1:])
if stage == "cluster":
    xyz, feat = tr["sa3"]["new_xyz"], tr["sa3"]["out"]
    new_xyz = tr["cluster"]["cand"]; idx = tr["cluster"]["ball_idx"][bi]
    scales, ns = cfg.cluster_scales, cfg.cluster_nsamples
    idxs, cnts = ops.ball_query_multi(scales, ns, xyz, new_xyz, tr["cluster"]["radius"], return_counts=True)
else:
    si = int(stage[2]) - 1
    xyz = pts[:, :, :3].contiguous() if si == 0 else tr[f"sa{si}"]["new_xyz"]
    feat = pts[:, :, 3:] if si == 0 else tr[f"sa{si}"]["out"]
    new_xyz = tr[stage]["new_xyz"]
    st = cfg.stages[si]
    idxs, cnts = ops.ball_query_multi(st.radii, st.nsamples, xyz, new_xyz, return_counts=True)
idx, cnt = idxs[bi], cnts[bi]
mlp = ops.PackedMLP(w[name], True, dev)
out = torch.zeros(idx.shape[0], idx.shape[1], mlp.out_channels, device=dev)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rows = int(cnt.clamp(min=1).sum().item())
print(f"{name}: groups {cnt.numel()} rows {rows} avg {rows/cnt.numel():.2f}")
for code in codes:
    mlp.default_geometry = code
    try:
        t = min(timeit(lambda: mlp.grouped(xyz, feat, new_xyz, idx, out=out, cnt=cnt)) for _ in range(3))
        print(f"  {code}: {t*1e3:.0f} us")
    except RuntimeError as e:
        print(f"  {code}: ERR {str(e)[-60:]}")
_lib.set_option("mlp_force", 0)
