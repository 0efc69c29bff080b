"""Times the bf16 MFMA chains (dense rows) on the KITTI topology shapes, B = 32, next to the f32 path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth
dev = torch.device("cuda:0")
B = 32
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rng = np.random.default_rng(0)
cases = [  # name, N, M, S, C, mlp
    ("sa1.b2", 16384, 4096, 64, 1, [32, 32, 64]),
    ("sa2.b2", 4096, 1024, 64, 64, [64, 96, 128]),
    ("sa3.b1", 1024, 512, 32, 128, [128, 192, 256]),
    ("sa3.b2", 1024, 512, 32, 128, [128, 256, 256]),
    ("cluster.b0", 512, 256, 16, 256, [256, 256, 512]),
    ("cluster.b1", 512, 256, 32, 256, [256, 512, 1024]),
]
for name, N, M, S, C, mlp in cases:
    dims = [C + 3] + mlp
    layers = synth.make_mlp_weights(dims, rng)
    xyz = torch.rand(B, N, 3, device=dev)
    feat = torch.randn(B, N, C, device=dev)
    new_xyz = xyz[:, :M].contiguous()
    idx = torch.randint(0, N, (B, M, S), device=dev, dtype=torch.int32)
    m16 = ops.PackedMLPBf16(layers, True, dev)
    m32 = ops.PackedMLP(layers, True, dev)
    fb = feat.bfloat16()
    out = torch.zeros(B, M, mlp[-1], device=dev)
    t16 = timeit(lambda: m16.grouped(xyz, fb, new_xyz, idx, out=out))
    t32 = timeit(lambda: m32.grouped(xyz, feat, new_xyz, idx, out=out))
    fl = 2.0 * B * M * S * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    print(f"{name}: dense {fl/1e9:.1f} GF  bf16 {t16:.3f} ms = {fl/t16/1e9:.1f} TF   f32 {t32:.3f} ms = {fl/t32/1e9:.1f} TF", flush=True)
for name, rows, dims in (("sa3.agg", B * 512, [768, 256]), ("cluster.agg", B * 256, [1536, 512]), ("head", B * 256, [512, 256, 256, 10])):
    layers = synth.make_mlp_weights(dims, rng)
    x = torch.randn(rows, dims[0], device=dev)
    m16 = ops.PackedMLPBf16(layers, False, dev); m32 = ops.PackedMLP(layers, False, dev)
    xb = x.bfloat16()
    t16 = timeit(lambda: m16.rows(xb)); t32 = timeit(lambda: m32.rows(x))
    fl = 2.0 * rows * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    print(f"{name}: {fl/1e9:.1f} GF  bf16 {t16:.3f} ms = {fl/t16/1e9:.1f} TF   f32 {t32:.3f} ms = {fl/t32/1e9:.1f} TF", flush=True)
