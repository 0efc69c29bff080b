"""Do two FPS workgroups on one CU slow each other?  FPS 16384 -> 4096 on B = 32 ... 1024 scenes (one workgroup per scene,
256 CUs): ms per launch.  Equal times for 256 and 512 scenes = a second workgroup on the CU rides in the first one's gaps."""
import sys
sys.path.insert(0, ".")
import torch, sad_amd
from sad_amd import ops, synth
x32 = torch.from_numpy(synth.make_batch(0, 32)).cuda()[:, :, :3].contiguous()
for B in (32, 128, 256, 512, 768, 1024):
    x = x32.repeat(B // 32, 1, 1).contiguous()
    ops.fps(x, 4096); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.fps(x, 4096); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"B={B}: fps ms {min(ts):.3f}  ({min(ts) / B * 32:.3f} ms per 32 scenes of chip time at this fill)", flush=True)
