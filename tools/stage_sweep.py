"""Times the MERGED dispatch of one stage's branches (sad_mlp_chain_multi_f32, prescanned tables) for the given
geometry codes (one code for all branches).  usage: python tools/stage_sweep.py sa3 2 4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth, _lib
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
for kv in filter(None, os.environ.get("SAD_OPTS", "").split(",")):
    k, v = kv.split("="); _lib.set_option(k, int(v))
stage = sys.argv[1]; codes = [int(c) for c in sys.argv[2:]]
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=False)
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
nets = [ops.PackedMLP(w[f"{stage}.b{i}"], True, dev, name=f"{stage}.b{i}") for i in range(len(mlps))]
wss = ops.rowscan_multi(idxs, cnts, xyz.shape[1])
width = sum(m[-1] for m in mlps)
out = torch.zeros(idxs[0].shape[0], idxs[0].shape[1], width, device=dev)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for code in codes:
    for n in nets: n.default_geometry = code
    calls, off = [], 0
    for n, idx, cnt, ws, m in zip(nets, idxs, cnts, wss, mlps):
        calls.append((n, xyz, feat, new_xyz, idx, out, off, cnt, ws)); off += m[-1]
    try:
        t = min(timeit(lambda: ops.grouped_multi(calls)) for _ in range(3))
        def both():
            out.zero_(); ops.grouped_multi(calls)
        tz = min(timeit(lambda: out.zero_()) for _ in range(3))
        tb = min(timeit(both) for _ in range(3))
        big = torch.empty(64 << 20, device=dev)          # 256 MB: evicts L2 and most of the Infinity Cache
        def cold():
            big.fill_(1.0); ops.grouped_multi(calls)
        tf = min(timeit(lambda: big.fill_(1.0)) for _ in range(3))
        tc = min(timeit(cold) for _ in range(3))
        print(f"  {stage} merged, geometry {code}: {t*1e3:.0f} us; after a zero fill of the output {(tb - tz)*1e3:.0f} us; after a 256 MB fill {(tc - tf)*1e3:.0f} us")
    except RuntimeError as e:
        print(f"  {code}: ERR {str(e)[-70:]}")
