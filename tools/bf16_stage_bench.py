"""Times the merged bf16 dispatch of each stage's branches (sad_mlp_chain_multi_bf16, prescanned tables, geometry 2) on
the real KITTI-shaped batch, and the plain chains.  SAD_AMD_LIB selects the build (A/B of kernel variants):
    python tools/bf16_stage_bench.py [stages...]      (default: sa1 sa2 sa3 cluster plain)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth, _lib
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
for kv in filter(None, os.environ.get("SAD_OPTS", "").split(",")):
    k, v = kv.split("="); _lib.set_option(k, int(v))
stages = sys.argv[1:] or ["sa1", "sa2", "sa3", "cluster", "plain"]
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
B = int(os.environ.get("SAD_B", "32"))
pts = torch.from_numpy(synth.make_batch(0, B)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=False, dtype="bf16")
tr = {}
det(pts, tr); torch.cuda.synchronize()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tag = os.path.basename(os.environ.get("SAD_AMD_LIB", "libsad_amd.so"))
for stage in stages:
    if stage == "plain":
        for name, x in (("sa1.agg", tr["sa1"]["cat"] if "cat" in tr["sa1"] else None), ):
            pass
        K = cfg.n_cand
        chains = [("sa1.agg", det.stages[0].agg, B * 4096, 128), ("sa2.agg", det.stages[1].agg, B * 1024, 384),
                  ("sa3.agg", det.stages[2].agg, B * 512, 768), ("cand", det.cand_mlp, B * K, 256),
                  ("cluster.agg", det.cluster_agg, B * K, 1536), ("head", det.head, B * K, 512)]
        tot = 0.0
        for name, m, rows, cin in chains:
            x = torch.randn(rows, cin, device=dev)
            xb = x.bfloat16()
            t = min(timeit(lambda: m.rows(x)) for _ in range(3))
            tb = min(timeit(lambda: m.rows(xb)) for _ in range(3))
            fl = 2.0 * rows * sum(a * b for a, b in zip(m.dims[:-1], m.dims[1:]))
            tot += min(t, tb)
            print(f"[{tag}] {name}: rows {rows} dims {m.dims}: f32-in {t*1e3:.1f} us, bf16-in {tb*1e3:.1f} us = {fl/tb/1e9:.0f} TF")
        print(f"[{tag}] plain chains total {tot*1e3:.0f} us")
        continue
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
    nets = [ops.PackedMLPBf16(w[f"{stage}.b{i}"], True, dev, name=f"{stage}.b{i}") for i in range(len(mlps))]
    wss = ops.rowscan_multi(idxs, cnts, xyz.shape[1])
    width = sum(m[-1] for m in mlps)
    out = torch.zeros(idxs[0].shape[0], idxs[0].shape[1], width, device=dev)
    rows = [int(c.clamp(min=1).sum().item()) for c in cnts]
    flops = sum(r * 2 * sum(a * b for a, b in zip(n.dims[:-1], n.dims[1:])) for r, n in zip(rows, nets))
    for n in nets: n.default_geometry = 2
    calls, off = [], 0
    for n, idx, cnt, ws, m in zip(nets, idxs, cnts, wss, mlps):
        calls.append((n, xyz, feat, new_xyz, idx, out, off, cnt, ws)); off += m[-1]
    t = min(timeit(lambda: ops.grouped_multi(calls)) for _ in range(3))
    line = f"[{tag}] {stage} merged (rows {rows}): {t*1e3:.0f} us = {flops/t/1e9:.0f} TF"
    for bi, c in enumerate(calls):                       # each chain alone
        tt = min(timeit(lambda: ops.grouped_multi([c, ]) if False else c[0].grouped(*c[1:5], out=c[5], col_off=c[6], cnt=c[7], ws=c[8])) for _ in range(3))
        line += f"; b{bi} {tt*1e3:.0f}"
    print(line, flush=True)
