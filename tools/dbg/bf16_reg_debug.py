"""Where does the register-resident bf16 chain differ from the oracle? (debug helper)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle, sad_amd
from sad_amd import ops, synth
oracle.build()
dev = torch.device("cuda:0")
B, N, M, S, C, mlp, r = 2, 3000, 700, 32, 1, [16, 16, 32], 0.08
if len(sys.argv) > 1:
    B, N, M, S, C = [int(v) for v in sys.argv[1:6]]
    mlp = [int(v) for v in sys.argv[6:9]]
    r = float(sys.argv[9])
rng = np.random.default_rng(N + M + S + C + sum(mlp))
xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
new_xyz = np.ascontiguousarray(xyz[:, :M])
X, Cn = torch.from_numpy(xyz).to(dev), torch.from_numpy(new_xyz).to(dev)
feat, F = None, None
if C and C <= 13:
    pts = rng.uniform(0, 1, (B, N, 3 + C)).astype(np.float32)
    pts[:, :, 3:] = oracle.bf16_round(pts[:, :, 3:])
    feat = np.ascontiguousarray(pts[:, :, 3:])
    F = torch.from_numpy(pts).to(dev)[:, :, 3:]
elif C:
    feat = oracle.bf16_round(rng.normal(size=(B, N, C)).astype(np.float32))
    F = torch.from_numpy(feat).to(dev).bfloat16()
idxs, cnts = ops.ball_query_multi((r,), (S,), X, Cn, return_counts=True)
layers = synth.make_mlp_weights([C + 3] + mlp, rng)
want = oracle.sa_group_mlp_max_bf16(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
net = ops.PackedMLPBf16(layers, True, dev)
net.default_geometry = 2
got = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
scale = np.abs(want).max()
bad = np.abs(got - want) > 1e-2 * scale
cnt = cnts[0].clamp(min=1).cpu().numpy().reshape(-1)
start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
end = start + cnt - 1
g_bad = np.where(bad.reshape(B * M, -1).any(1))[0]
print(f"groups {B*M}, bad groups {len(g_bad)}, bad elements {bad.sum()} of {bad.size}")
for g in g_bad[:40]:
    ch = np.where(bad.reshape(B * M, -1)[g])[0]
    print(f" g={g} rows {start[g]}..{end[g]} (tile {start[g]//32}..{end[g]//32}, in-tile {start[g]%32}..{end[g]%32}) cnt {cnt[g]} bad ch {ch[:8]} n={len(ch)} got {got.reshape(B*M,-1)[g, ch[:3]]} want {want.reshape(B*M,-1)[g, ch[:3]]}")
# statistics: are bad groups those crossing a 16-row boundary?
cross16 = (start // 16) != (end // 16)
print("bad & cross16:", int(cross16[g_bad].sum()), " bad & not cross16:", int((~cross16[g_bad]).sum()), " total cross16:", int(cross16.sum()))
