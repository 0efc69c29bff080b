"""How much does one FPS kernel (32 workgroups of 1024 threads, 2.8 ms) running on another stream slow a stage's
MLP dispatch down?  Times the SA3 / cluster / SA2 / SA1 dispatches alone and under 1 or 2 concurrent FPS kernels."""
import os, sys, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
det.set_geometry(json.load(open("profiles/r03_geometry.json")))
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
sides = [torch.cuda.Stream(), torch.cuda.Stream()]
main = torch.cuda.Stream()
def timed(calls, nfps, reps=4):
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        for s in sides[:nfps]:
            with torch.cuda.stream(s):
                for _ in range(2):
                    ops.fps(xyz0, 4096)            # 2 x 2.9 ms per side stream: covers the timed dispatches
        with torch.cuda.stream(main):
            ops.grouped_multi(calls)               # (starts while the FPS kernels run)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            for _ in range(reps):
                ops.grouped_multi(calls)
            e1.record(main)
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e3
        best = t if best is None or t < best else best
    return best
for stage in os.environ.get("SAD_PROBE_STAGES", "sa1,sa2,sa3,cluster").split(","):
    calls = stage_calls(stage)
    print(f"{stage:8s} alone {timed(calls, 0):6.0f} us   under 1 FPS kernel {timed(calls, 1):6.0f} us   under 2 {timed(calls, 2):6.0f} us")
