"""Times the FPS kernel variants at the three stage sizes of the KITTI topology (B=32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import ops, synth, _lib
dev = torch.device("cuda:0")
B = 32
pts = torch.from_numpy(synth.make_batch(0, B)).to(dev)
xyz = pts[:, :, :3].contiguous()
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cur = xyz
GEOS = {16384: (832, 1616), 4096: (416, 808, 1604, 232, 816), 1024: (404, 208, 116, 408, 804)}
for N, M in ((16384, 4096), (4096, 1024), (1024, 512)):
    ref = None
    variants = [("key", 2, 0), ("wavebucket", 3, 1024)] + [(f"cell{g}", 4, g) for g in GEOS[N]]
    for name, var, th in variants:
        _lib.set_option("fps_variant", var); _lib.set_option("fps_threads", th)
        t = timeit(lambda: ops.fps(cur, M))
        idx = ops.fps(cur, M)
        if ref is None: ref = idx
        same = bool(torch.equal(ref, idx))
        print(f"N={N} M={M} {name}: {t:.3f} ms  {1e3*t/M:.3f} us/step same={same}", flush=True)
    cur = ops.gather_xyz(cur, ref)
