"""Drop-in generality: sa_module chains that are NOT among the compiled shapes of the register-resident /
cooperative kernels, on real grouped rows (KITTI-shaped scenes: FPS centroids, ball query with counts, row-packing scan), f32:
ms per dispatch, executed TFLOP/s and fraction of the 157.3 TFLOP/s f32 MFMA peak, for the kernel the library prefers un-tuned
and for the autotuned pick.  usage: python tools/generality_bench.py [B=32] [dense|kitti]   (dense scenes: enough rows per dispatch for the fraction to mean something)"""
import os, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sad_amd, numpy as np, torch
from sad_amd import ops, synth, _lib
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PEAK = 157.3
SCENE = sys.argv[2] if len(sys.argv) > 2 else "dense"
pts = torch.from_numpy((synth.make_dense_batch if SCENE == "dense" else synth.make_batch)(0, B, 16384)).to(dev)      # dense: 20 m x 20 m, full neighbourhoods
xyz0 = pts[:, :, :3].contiguous()
idx1 = ops.fps(xyz0, 4096)
xyz1 = ops.gather_xyz(xyz0, idx1)                      # 4096 centroids (SA1 level)
xyz2 = xyz1[:, :1024].contiguous()                     # nested prefixes = FPS of the level above
xyz3 = xyz1[:, :512].contiguous()
g = torch.Generator(device="cpu").manual_seed(5)

def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        best = ms if best is None or ms < best else best
    return best

def chain_flops(dims):
    return 2 * sum(a * b for a, b in zip(dims[:-1], dims[1:]))

cases = [  # name, source level xyz, centroids, radius, nsample, C (feature channels), mlp
    ("ssg-like SA1: C=6 [32,32,64] r=0.2 S=32", xyz0, xyz1, 0.2, 32, 6, [32, 32, 64]),
    ("C=32 [64,96,128] r=0.4 S=32", xyz1, xyz2, 0.4, 32, 32, [64, 96, 128]),
    ("C=64 [64,128] (2 layers) r=0.8 S=64", xyz1, xyz2, 0.8, 64, 64, [64, 128]),
    ("C=96 [128,196,256] r=1.2 S=32", xyz2, xyz3, 1.2, 32, 96, [128, 196, 256]),
    ("C=128 [128,128,256,256] (4 layers) r=1.2 S=32", xyz2, xyz3, 1.2, 32, 128, [128, 128, 256, 256]),
    ("C=256 [256,384,512] r=1.6 S=32", xyz2, xyz3, 1.6, 32, 256, [256, 384, 512]),
    ("compiled control: C=128 [128,128,256] r=1.2 S=32 (KITTI sa3.b0)", xyz2, xyz3, 1.2, 32, 128, [128, 128, 256]),
]
rng = np.random.default_rng(0)
rows_out = []
for name, src, cen, r, S, C, mlp in cases:
    N, M = src.shape[1], cen.shape[1]
    dims = [C + 3] + mlp
    layers = synth.make_mlp_weights(dims, rng)
    net = ops.PackedMLP(layers, True, dev, name=name)
    feat = torch.randn((B, N, C), generator=g).to(dev) if C else None
    (idx,), (cnt,) = ops.ball_query_multi([r], [S], src, cen, return_counts=True)
    rows = int(cnt.sum().item())
    gf = rows * chain_flops(dims) / 1e9
    out = torch.zeros((B, M, mlp[-1]), device=dev)
    res = {}
    pref = net.preferred_geometry
    for label in ("untuned", "autotuned"):
        if label == "autotuned":
            ops.AUTOTUNE = True
            try:
                net.grouped(src, feat, cen, idx, out=out, cnt=cnt)
                torch.cuda.synchronize()
            finally:
                ops.AUTOTUNE = False
        out.zero_()
        ms = timeit(lambda: net.grouped(src, feat, cen, idx, out=out, cnt=cnt))
        code = (list(net._geom.values())[-1] if net._geom else 0) if label == "autotuned" else pref
        res[label] = (ms, code)
    line = {"chain": name, "dims": dims, "packed_as": net.pack_dims if getattr(net, "padded", False) else None, "rows": rows, "row_fraction": round(rows / (B * M * S), 3), "gflop": round(gf, 2), "preferred_geometry": pref}
    for label, (ms, code) in res.items():
        line[label] = {"ms": round(ms, 4), "tflops": round(gf / ms, 1), "frac": round(gf / ms / PEAK, 3), "geometry": int(code)}
    rows_out.append(line)
    print(json.dumps(line), flush=True)
# plain-row chain (no grouping): 8192 and 131072 rows
for nrows in (8192, 131072):
    dims = [256, 512, 1024]
    net = ops.PackedMLP(synth.make_mlp_weights(dims, rng), False, dev, name="plain")
    x = torch.randn((nrows, 256), generator=g).to(dev)
    ops.AUTOTUNE = True
    try:
        net.rows(x); torch.cuda.synchronize()
    finally:
        ops.AUTOTUNE = False
    ms = timeit(lambda: net.rows(x))
    gf = nrows * chain_flops(dims) / 1e9
    print(json.dumps({"chain": f"plain rows {dims} x {nrows}", "gflop": round(gf, 2), "autotuned": {"ms": round(ms, 4), "tflops": round(gf / ms, 1), "frac": round(gf / ms / PEAK, 3), "geometry": int(list(net._geom.values())[-1])}}), flush=True)
