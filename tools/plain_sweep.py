"""Times geometry codes for a plain-row chain.  usage: python tools/plain_sweep.py ROWS C0,C1[,C2..] code code ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import ops, synth, _lib
dev = torch.device("cuda:0")
rows = int(sys.argv[1]); dims = [int(v) for v in sys.argv[2].split(",")]
codes = [int(c) for c in sys.argv[3:]] or [0]
rng = np.random.default_rng(0)
net = ops.PackedMLP(synth.make_mlp_weights(dims, rng), False, dev)
x = torch.from_numpy(np.maximum(rng.normal(size=(rows, dims[0])).astype(np.float32), 0)).to(dev)
out = torch.empty((rows, dims[-1]), device=dev)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gf = 2.0 * rows * sum(a * b for a, b in zip(dims[:-1], dims[1:])) / 1e9
for code in codes:
    net.default_geometry = code
    try:
        t = min(timeit(lambda: net.rows(x, out=out)) for _ in range(3))
        print(f"  {dims} rows {rows} geometry {code}: {t*1e3:.1f} us  {gf / t:.1f} TFLOP/s")
    except RuntimeError as e:
        print(f"  {code}: ERR {str(e)[-60:]}")
