"""Times chosen geometry codes for the SA1 grouped chains on the real ball-query output (B=32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth, _lib
dev = torch.device("cuda:0")
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
xyz = pts[:, :, :3].contiguous(); feat = pts[:, :, 3:]
st = cfg.stages[0]
new_xyz = ops.gather_xyz(xyz, ops.fps(xyz, st.npoint))
idxs, cnts = ops.ball_query_multi(st.radii, st.nsamples, xyz, new_xyz, return_counts=True)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for bi in range(3):
    mlp = ops.PackedMLP(w[f"sa1.b{bi}"], True, dev)
    out = torch.zeros(32, st.npoint, mlp.out_channels, device=dev)
    ref = None
    line = [f"sa1.b{bi} avg cnt {cnts[bi].float().mean().item():.2f}:"]
    for code in (0, 801, 802, 811, 1, 2001, 3001, 4001, 5001, 6001, 7001, 4801, 5801, 6801):
        _lib.set_option("mlp_force", code)
        try:
            t = timeit(lambda: mlp.grouped(xyz, feat, new_xyz, idxs[bi], out=out, cnt=cnts[bi]))
        except RuntimeError as e:
            line.append(f"{code}:ERR"); continue
        out.zero_(); mlp.grouped(xyz, feat, new_xyz, idxs[bi], out=out, cnt=cnts[bi])
        if ref is None: ref = out.clone()
        line.append(f"{code}:{t*1e3:.0f}us{'' if torch.equal(ref, out) else '(DIFF)'}")
    _lib.set_option("mlp_force", 0)
    print(" ".join(line), flush=True)
