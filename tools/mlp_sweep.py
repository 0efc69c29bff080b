"""Times every workgroup geometry (W, WN, RW) of mlp_chain_kernel for each MLP launch of the KITTI
topology at bench size (B=32).  Run on the GPU box:  python tools/mlp_sweep.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth, _lib

dev = torch.device("cuda:0")
cfg = config.KITTI
B = int(os.environ.get("B", "32"))
w = synth.make_weights(cfg, 0)
rng = np.random.default_rng(0)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

cases = []  # (name, grouped, N, M, S, C)
n, c = cfg.n_points, cfg.in_feat
for si, st in enumerate(cfg.stages):
    for bi, (s, mlp) in enumerate(zip(st.nsamples, st.mlps)):
        cases.append((f"sa{si+1}.b{bi}", True, n, st.npoint, s, c))
    cat = sum(m[-1] for m in st.mlps)
    cases.append((f"sa{si+1}.agg", False, 0, st.npoint, 1, cat))
    n, c = st.npoint, st.agg
for bi, s in enumerate(cfg.cluster_nsamples):
    cases.append((f"cluster.b{bi}", True, n, cfg.n_cand, s, c))
cases.append(("cluster.agg", False, 0, cfg.n_cand, 1, sum(m[-1] for m in cfg.cluster_mlps)))
cases.append(("head", False, 0, cfg.n_cand, 1, cfg.cluster_agg))
only = os.environ.get("ONLY")
res = {}
for name, grouped, N, M, S, C in cases:
    if only and only not in name: continue
    layers = w[name]
    relu = None
    if name == "head": relu = 0b011
    mlp = ops.PackedMLP(layers, grouped, dev, relu_mask=relu, name=name)
    if grouped:
        xyz = torch.rand(B, N, 3, device=dev)
        new_xyz = torch.rand(B, M, 3, device=dev)
        idx = torch.randint(0, N, (B, M, S), device=dev, dtype=torch.int32)
        feat = torch.randn(B, N, C, device=dev) if C > 1 else torch.rand(B, N, 4, device=dev)[:, :, 3:]
        out = torch.empty(B, M, mlp.out_channels, device=dev)
        fn = lambda: mlp.grouped(xyz, feat, new_xyz, idx, out=out)
        rows = B * M * S
    else:
        x = torch.randn(B * M, C, device=dev)
        out = torch.empty(B * M, mlp.out_channels, device=dev)
        fn = lambda: mlp.rows(x, out=out)
        rows = B * M
    dims = mlp.dims
    flops = 2 * rows * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    _lib.set_option("mlp_force", 0)
    base = timeit(fn)
    line = [f"{name:12s} dims={dims} rows={rows} default={base:.3f}ms {flops/base/1e9:.1f}TF |"]
    best = (base, "default")
    for W in (4, 8):
        for wns in range(0, 4):
            if (1 << wns) > W: continue
            for rw in (1, 2, 4):
                code = W * 100 + wns * 10 + rw
                _lib.set_option("mlp_force", code)
                try:
                    t = timeit(fn, 3)
                except RuntimeError:
                    continue
                line.append(f"{code}:{t:.3f}")
                if t < best[0]: best = (t, code)
    _lib.set_option("mlp_force", 0)
    res[name] = best
    print(" ".join(line), "| best", best, f"{flops/best[0]/1e9:.1f}TF", flush=True)
print(json.dumps({k: v[1] for k, v in res.items()}))
print("sum best ms", sum(v[0] for v in res.values()))
