import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle, sad_amd
from sad_amd import ops, synth, _lib
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
B, N, M, S, C = 1, 1024, 64, 32, 0
mlp = [64, 64, 128]
xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
new_xyz = np.ascontiguousarray(xyz[:, :M])
X, Cn = torch.from_numpy(xyz).to(dev), torch.from_numpy(new_xyz).to(dev)
idxs, cnts = ops.ball_query_multi((0.2,), (S,), X, Cn, return_counts=True)
layers = synth.make_mlp_weights([C + 3] + mlp, rng)
want = oracle.sa_group_mlp_max(xyz, None, new_xyz, idxs[0].cpu().numpy(), layers)
net = ops.PackedMLP(layers, True, dev)
_lib.set_option("mlp_force", 2)
got = net.grouped(X, None, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
_lib.set_option("mlp_force", 0)
bad = got != want
print("wrong fraction", bad.mean(), "cnt:", cnts[0][0, :16].cpu().numpy())
print("per-channel wrong frac (first 40):", np.round(bad[0].mean(0)[:40], 2))
print("per-group wrong frac (first 40):", np.round(bad[0].mean(1)[:40], 2))
print("got[0,0,:8]", got[0, 0, :8], "\nwant[0,0,:8]", want[0, 0, :8])
print("got[0,5,:8]", got[0, 5, :8], "\nwant[0,5,:8]", want[0, 5, :8])
# is got a max over a subset / superset?  compare with per-row outputs
idx = idxs[0].cpu().numpy()[0]
j = idx.reshape(-1)
rows = (xyz[0][j] - np.repeat(new_xyz[0], S, 0)).astype(np.float32)
y = oracle.mlp_rows(rows, layers).reshape(M, S, -1)
for g in range(4):
    c = int(cnts[0][0, g])
    print("group", g, "cnt", c, "got ch0", got[0, g, 0], "row values ch0:", np.round(y[g, :c, 0], 4))
