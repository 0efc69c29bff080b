"""Phase breakdown of one FPS step (measurement build: bash tools/probe/build_stamps.sh fpsst fps_bucket.hip -DSAD_FPS_STAMPS;
SAD_AMD_LIB=build/libsad_fpsst.so python tools/probe/fps_stamps.py)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sad_amd
from sad_amd import ops, synth, _lib
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.make_batch(0, 32)).to(dev)[:, :, :3].contiguous()
ops.fps(x, 4096); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.fps(x, 4096); e1.record(); torch.cuda.synchronize()
print(f"fps: {e0.elapsed_time(e1):.3f} ms for 4095 steps = {e0.elapsed_time(e1) / 4095 * 1e3:.3f} us per step")
buf = (ctypes.c_ulonglong * 128)()
L = _lib.lib()
L.sad_debug_read_fps_stamps.restype = ctypes.c_int
assert L.sad_debug_read_fps_stamps(buf) == 0
a = np.array(buf[:], dtype=np.float64).reshape(16, 8)
steps = a[:, 7]
print("s_memtime ticks (~ shader cycles) per step, per wave of scene 0, second half of the steps; the stamps themselves add ~40 % to a step")
print("wave  skip  update  publish  barrier  read   | active steps  buckets/active step")
for w in range(16):
    r = a[w]
    print(f"{w:4d} {r[0]/steps[w]:5.1f} {r[1]/steps[w]:7.1f} {r[2]/steps[w]:8.1f} {r[3]/steps[w]:8.1f} {r[4]/steps[w]:6.1f}   | {r[5]/steps[w]:6.2f}  {r[6]/max(1.0, r[5]):5.2f}")
tot = a[:, :5].sum(1) / steps
print("sum per wave:", np.round(tot, 1))
