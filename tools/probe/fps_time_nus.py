import os, sys
sys.path.insert(0, ".")
import torch, sad_amd
from sad_amd import ops, synth
x = torch.from_numpy(synth.make_nuscenes_batch(0, 8)).cuda()[:, :, :3].contiguous()
ops.fps(x, 16384); torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.fps(x, 16384); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(os.environ.get("SAD_AMD_LIB", "tree"), "fps 65536 -> 16384, 8 scenes, ms:", min(ts))
