"""The f32 row-streaming layer (geometry 5) alone, first form against second (sad_set_option mlp_rows_form), back to back launches, and both
checked against each other bit for bit:  python tools/probe/rows_f32_forms_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sad_amd
from sad_amd import _lib, ops, synth
dev = torch.device("cuda:0")
shapes = [("sa1.agg", 131072, 128, 64), ("sa2.agg", 32768, 384, 128), ("sa3.agg", 16384, 768, 256), ("cluster.agg", 8192, 1536, 512)]
rng = np.random.default_rng(0)
for name, R, K, CO in shapes:
    mlp = ops.PackedMLP(synth.make_mlp_weights([K, CO], rng), False, dev)
    mlp.default_geometry = 5
    x = torch.randn((R, K), device=dev)
    outs, res = [], []
    for form in (1, 0):
        _lib.set_option("mlp_rows_form", form)
        out = torch.empty((R, CO), device=dev)
        for _ in range(3): mlp.rows(x, out=out)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): mlp.rows(x, out=out)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        res.append(best); outs.append(out)
    _lib.set_option("mlp_rows_form", 0)
    print(f"{name:12s} {R:7d} x {K:5d} -> {CO:4d}: first form {res[0]:6.1f} us, second {res[1]:6.1f} us ({2 * R * K * CO / res[1] / 1e6:.0f} TFLOP/s = {2 * R * K * CO / res[1] / 1e6 / 157.3:.2f} of the f32 peak); bit-equal: {torch.equal(outs[0], outs[1])}")
