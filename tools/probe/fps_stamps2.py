"""Phase breakdown of the critical wave's step, second form of the cell kernel (measurement build:
bash tools/probe/build_stamps.sh fpsst2 fps_bucket.hip -DSAD_FPS_STAMPS2;  SAD_AMD_LIB=build/libsad_fpsst2.so python tools/probe/fps_stamps2.py [geometry])."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sad_amd
from sad_amd import ops, synth, _lib
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.make_batch(0, 32)).to(dev)[:, :, :3].contiguous()
_lib.set_option("fps_variant", 0)      # the second form (fps_cell2_kernel): the only one that writes g_fpst2; 6 = first form
if len(sys.argv) > 1: _lib.set_option("fps_threads", int(sys.argv[1]))
ops.fps(x, 4096); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.fps(x, 4096); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"fps (stamped build): {ms:.3f} ms for 4095 steps = {ms / 4095 * 1e3:.3f} us per step")
buf = (ctypes.c_ulonglong * 128)()
L = _lib.lib()
L.sad_debug_read_fps_stamps2.restype = ctypes.c_int
assert L.sad_debug_read_fps_stamps2(buf) == 0
a = np.array(buf[:], dtype=np.float64).reshape(16, 8)
n = a[:, 6].sum()
assert n > 0, "no stamps: the library was not built with -DSAD_FPS_STAMPS2, or fps_variant selects the first form"
tot = a.sum(0)
names = ["barrier->centre", "skip test", "bucket updates", "wave best", "record", "key atomic->barrier passed"]
print(f"steps in which the stamped wave held the sampled point: {int(n)} (all waves of scene 0, second half of the run); buckets updated per such step {tot[7] / n:.2f}")
s = 0
for i, nm in enumerate(names):
    print(f"  {nm:28s} {tot[i] / n:8.1f} ticks"); s += tot[i] / n
print(f"  {'sum':28s} {s:8.1f} ticks per critical step;  wall per step {ms / 4095 * 1e6:.0f} ns -> {ms / 4095 * 1e6 / s:.2f} ns per tick")
