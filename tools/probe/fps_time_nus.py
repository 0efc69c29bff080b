"""FPS alone, 32 nuScenes-shaped scenes 65536 -> 16384: ms per batch for the fps_variant values given (default 0; 6 = first form)."""
import os, sys
sys.path.insert(0, ".")
import torch, sad_amd
from sad_amd import ops, synth, _lib
x = torch.from_numpy(synth.make_nuscenes_batch(0, 32)).cuda()[:, :, :3].contiguous()
ref = None
for v in [int(a) for a in sys.argv[1:]] or [0]:
    _lib.set_option("fps_variant", v)
    out = ops.fps(x, 16384); torch.cuda.synchronize()
    if ref is None: ref = out
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.fps(x, 16384); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(os.environ.get("SAD_AMD_LIB", "tree"), f"variant {v}: fps 65536 -> 16384, 32 scenes, ms {min(ts):.3f}  us/step {min(ts) / 16.383:.3f}  same as first: {bool(torch.equal(out, ref))}", flush=True)
_lib.set_option("fps_variant", 0)
