import os, sys
sys.path.insert(0, ".")
import torch, sad_amd
from sad_amd import ops, synth
x = torch.from_numpy(synth.make_batch(0, 32)).cuda()[:, :, :3].contiguous()
ops.fps(x, 4096); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.fps(x, 4096); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(os.environ.get("SAD_AMD_LIB", "tree"), "fps ms:", min(ts))
