"""Phase breakdown of a round of the third form of the cell kernel (several samples per round), per wave of workgroup 0:
bash tools/probe/build_variant.sh fpsst3 "-DSAD_FPS_STAMPS3" fps_bucket.hip;  SAD_AMD_LIB=build/libsad_fpsst3.so python tools/probe/fps_stamps3.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sad_amd, numpy as np, torch
from sad_amd import ops, synth, _lib
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.make_batch(0, 32)).to(dev)[:, :, :3].contiguous()
_lib.set_option("fps_variant", 7)
if len(sys.argv) > 1: _lib.set_option("fps_threads", int(sys.argv[1]))
ops.fps(x, 4096); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.fps(x, 4096); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
buf = (ctypes.c_ulonglong * 192)()
L = _lib.lib()
assert L.sad_debug_read_fps_stamps3(buf) == 0
a = np.array(buf[:], dtype=np.float64).reshape(16, 12)
rounds = a[0, 7]
assert rounds > 0, "no stamps: build with -DSAD_FPS_STAMPS3"
print(f"fps (stamped build): {ms:.3f} ms; second half: {int(rounds)} rounds, {a[0, 10] / rounds:.2f} samples per round")
names = ["box tests", "bucket updates", "top-2 + publish", "wait barrier 1", "ranking", "wait barrier 2", "prefix + samples"]
print("ticks per round, mean over waves / max over waves / wave 0:")
tot = 0
for i, n in enumerate(names):
    v = a[:, i] / rounds
    print(f"  {n:18s} {v.mean():8.1f} {v.max():8.1f} {v[0]:8.1f}"); tot += v.mean()
print(f"  {'sum':18s} {tot:8.1f} ticks per round;  wall per round {ms * 1e6 / (2 * rounds + 1):.0f} ns (whole run / rounds, approx)")
print(f"buckets updated per wave and round {a[:, 8].mean() / rounds:.2f} (max wave {a[:, 8].max() / rounds:.2f}); rounds with a top-2 recompute per wave {a[:, 9].mean() / rounds:.2f}")
