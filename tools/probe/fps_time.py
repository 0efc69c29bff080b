"""FPS alone, 32 KITTI-shaped scenes 16384 -> 4096: ms per batch for the fps_variant values given (default 0)."""
import os, sys
sys.path.insert(0, ".")
import torch, sad_amd
from sad_amd import ops, synth, _lib
x = torch.from_numpy(synth.make_batch(0, 32)).cuda()[:, :, :3].contiguous()
ref = None
for v in [int(a) for a in sys.argv[1:]] or [0]:
    geo = 0
    if v >= 100: geo, v = v // 10, v % 10          # e.g. 8326 = geometry 832, variant 6
    _lib.set_option("fps_variant", v); _lib.set_option("fps_threads", geo)
    out = ops.fps(x, 4096); torch.cuda.synchronize()
    if ref is None: ref = out
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.fps(x, 4096); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(os.environ.get("SAD_AMD_LIB", "tree"), f"variant {v} geometry {geo or 'auto'}: fps ms {min(ts):.3f}  us/step {min(ts) / 4.095:.3f}  same as first: {bool(torch.equal(out, ref))}", flush=True)
_lib.set_option("fps_variant", 0); _lib.set_option("fps_threads", 0)
