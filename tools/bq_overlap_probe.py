"""Do the SA1 ball-query kernels (grid build + grid query, 3 radii, 32 scenes: ~140 us alone) overlap with a stage's MLP
dispatch running on another stream, or do the two time-share the chip?  Prints both alone, and the time for BOTH when
launched together on two streams (perfect overlap = max, time-sharing = sum)."""
import os, sys, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
xyz0 = pts[:, :, :3].contiguous()
def stage_calls(stage):
    if stage == "cluster":
        xyz, feat, new_xyz = tr["sa3"]["new_xyz"], tr["sa3"]["out"], tr["cluster"]["cand"]
        idxs, cnts = ops.ball_query_multi(cfg.cluster_scales, cfg.cluster_nsamples, xyz, new_xyz, tr["cluster"]["radius"], return_counts=True)
        nets, mlps = det.cluster_branches, cfg.cluster_mlps
    else:
        si = int(stage[2]) - 1
        xyz = xyz0 if si == 0 else tr[f"sa{si}"]["new_xyz"]
        feat = pts[:, :, 3:] if si == 0 else tr[f"sa{si}"]["out"]
        new_xyz = tr[stage]["new_xyz"]; st = cfg.stages[si]
        idxs, cnts = ops.ball_query_multi(st.radii, st.nsamples, xyz, new_xyz, return_counts=True)
        nets, mlps = det.stages[si].branches, st.mlps
    wss = ops.rowscan_multi(idxs, cnts, xyz.shape[1])
    out = torch.zeros(idxs[0].shape[0], idxs[0].shape[1], sum(m[-1] for m in mlps), device=dev)
    calls, off = [], 0
    for n, idx, cnt, ws, m in zip(nets, idxs, cnts, wss, mlps):
        calls.append((n, xyz, feat, new_xyz, idx, out, off, cnt, ws)); off += m[-1]
    return calls
st1 = cfg.stages[0]
new1 = tr["sa1"]["new_xyz"]
def query():
    ops.ball_query_multi(st1.radii, st1.nsamples, xyz0, new1, return_counts=True)
side, main = torch.cuda.Stream(), torch.cuda.Stream()
def both(calls, nq, nm, do_q=True, do_m=True):
    best = None
    for _ in range(4):
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        side.wait_event(e0); main.wait_event(e0)
        if do_q:
            with torch.cuda.stream(side):
                for _ in range(nq): query()
                e1.record(side)
        if do_m:
            with torch.cuda.stream(main):
                for _ in range(nm): ops.grouped_multi(calls)
                e2.record(main)
        torch.cuda.synchronize()
        t = max(e0.elapsed_time(e1) if do_q else 0.0, e0.elapsed_time(e2) if do_m else 0.0) * 1e3
        best = t if best is None or t < best else best
    return best
for stage, nm in (("sa3", 2), ("cluster", 2)):
    calls = stage_calls(stage)
    nq = 8
    tq = both(calls, nq, nm, True, False); tm = both(calls, nq, nm, False, True); tb = both(calls, nq, nm, True, True)
    print(f"{stage}: {nq} SA1 queries alone {tq:6.0f} us, {nm} dispatches alone {tm:6.0f} us, together {tb:6.0f} us  (sum {tq + tm:6.0f}, max {max(tq, tm):6.0f})")
