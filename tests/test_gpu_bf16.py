"""GPU parity of the bf16 MFMA grouped-MLP path (SPEC.md §14, BASELINE.json configs[4]) (-m gpu).

Tolerance (SPEC §14): the binary32 accumulation order inside the matrix core is unspecified and a
hidden activation may round to the neighbouring bfloat16, so
    |gpu - oracle| <= 1e-2 * max|oracle|   element-wise,   mean|gpu - oracle| <= 1e-5 * max|oracle|.
The oracle (oracle.mlp_rows_bf16) rounds inputs/weights/activations to bf16 and sums in binary64.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MAX_TOL, MEAN_TOL = 1e-2, 1e-5


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _close(got, want, what):
    scale = max(float(np.abs(want).max()), 1e-6)
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    print(f"[parity-bf16] {what}: max|diff|/scale={diff.max() / scale:.3e} mean={diff.mean() / scale:.3e}")
    assert diff.max() <= MAX_TOL * scale, f"{what}: max diff {diff.max() / scale:.3e} of scale"
    assert diff.mean() <= MEAN_TOL * scale, f"{what}: mean diff {diff.mean() / scale:.3e} of scale"


PLAIN = [
    (37, [7, 12, 5], None),
    (1000, [128, 64], None),
    (257, [768, 256], None),
    (130, [1536, 512], None),
    (256, [512, 256, 256, 10], 0b011),
    (64, [33, 40, 50, 60, 70], None),
]


@pytest.mark.parametrize("rows,dims,mask", PLAIN)
@pytest.mark.parametrize("in_bf16", [True, False])
def test_rows_bf16(orc, sad, dev, rows, dims, mask, in_bf16):
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(sum(dims) + rows)
    layers = synth.make_mlp_weights(dims, rng)
    x = rng.normal(size=(rows, dims[0])).astype(np.float32)
    mlp = ops.PackedMLPBf16(layers, False, dev, relu_mask=mask)
    xt = _t(x, dev)
    got = mlp.rows(xt.bfloat16() if in_bf16 else xt).cpu().numpy()
    want = orc.mlp_rows_bf16(x, layers, relu_mask=mask)
    _close(got, want, f"rows {dims} in_bf16={in_bf16}")
    # bf16 output into a slice of a wider buffer
    buf = torch.full((rows, dims[-1] + 12), -7.0, device=dev, dtype=torch.bfloat16)
    mlp.rows(xt, out=buf, col_off=8)
    b = buf.float().cpu().numpy()
    _close(b[:, 8:8 + dims[-1]], orc.bf16_round(want), "bf16 slice")
    assert (b[:, :8] == -7).all() and (b[:, 8 + dims[-1]:] == -7).all()


# The row-streaming layer has two chunk loops (csrc/mlp_bf16_rows.hip, launch_bf16_rows): rows and weights queued two chunks ahead for layers
# of more than two chunks on at most two workgroups per CU, the single-ahead loop otherwise.  PLAIN above runs its deep layers on tiny grids
# (queued loop); 32 nuScenes-shaped scenes put the same layers on ~1 024 workgroups (single-ahead loop).  These cases are the second kind.
BIG_GRID = [
    (66_000, [384, 128]),       # 6 chunks, one channel block, ragged last row block
    (33_000, [768, 256]),       # 12 chunks, two channel blocks
]


@pytest.mark.parametrize("rows,dims", BIG_GRID)
def test_rows_bf16_deep_layer_on_a_large_grid(orc, sad, dev, rows, dims):
    import torch
    from sad_amd import ops, synth
    # the launcher's grid: 8 * ceil(row blocks / 8) * channel blocks (128 rows and up to 4 x 32 channels per workgroup)
    ct = (dims[-1] + 31) // 32
    ncb = (ct + 3) // 4
    grid = 8 * ((((rows + 127) // 128) + 7) // 8) * ncb
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    assert (dims[0] + 15) // 16 > 8 and grid > 2 * cus, f"this case no longer reaches the single-ahead loop of a deep layer (grid {grid}, {cus} CUs)"
    rng = np.random.default_rng(rows + sum(dims))
    layers = synth.make_mlp_weights(dims, rng)
    x = rng.normal(size=(rows, dims[0])).astype(np.float32)
    mlp = ops.PackedMLPBf16(layers, False, dev)
    got = mlp.rows(_t(x, dev)).cpu().numpy()
    want = orc.mlp_rows_bf16(x, layers)
    _close(got, want, f"rows {dims} x {rows} (grid {grid} on {cus} CUs)")
    # the same rows on a small grid take the queued loop: both loops see the same operands in the same k order per chunk
    few = 1000
    got_few = mlp.rows(_t(x[:few], dev)).cpu().numpy()
    _close(got_few, want[:few], f"rows {dims} x {few} (queued loop)")


GROUPED = [
    # (B, N, M, S, C, mlp, radius)
    (1, 1024, 256, 32, 0, [64, 64, 128], 0.2),
    (2, 2048, 512, 32, 1, [16, 16, 32], 0.15),
    (2, 1024, 128, 64, 64, [64, 96, 128], 0.3),
    (1, 512, 100, 16, 256, [256, 256, 512], 0.4),
    (1, 512, 64, 32, 256, [256, 512, 1024], 0.5),
    (1, 700, 33, 24, 8, [32, 48], 0.3),              # nsample not a power of two: generic pooling
    (1, 300, 7, 16, 5, [24, 40], 0.5),               # odd group count, feature width not a multiple of 8
]


@pytest.mark.parametrize("B,N,M,S,C,mlp,radius", GROUPED)
def test_grouped_bf16(orc, sad, dev, B, N, M, S, C, mlp, radius):
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(N + M + S + C)
    xyz = rng.random((B, N, 3), dtype=np.float32)
    feat = orc.bf16_round(rng.normal(size=(B, N, C)).astype(np.float32)) if C else None
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    new_xyz = np.stack([xyz[b][orc.fps(xyz[b:b + 1], M)[0]] for b in range(B)])
    idx = orc.ball_query(radius, S, xyz, new_xyz)
    want = orc.sa_group_mlp_max_bf16(xyz, feat, new_xyz, idx, layers)
    m = ops.PackedMLPBf16(layers, True, dev)
    ft = _t(feat, dev).bfloat16() if C else None
    got = m.grouped(_t(xyz, dev), ft, _t(new_xyz, dev), _t(idx, dev)).cpu().numpy()
    _close(got, want, f"grouped N={N} M={M} S={S} C={C} {mlp}")
    if C:   # float32 features are rounded on load: same result
        got32 = m.grouped(_t(xyz, dev), _t(feat, dev), _t(new_xyz, dev), _t(idx, dev)).cpu().numpy()
        assert np.array_equal(got32, got)
    # padding rows skipped (counts from the ball query): a duplicate row cannot change the max, and a
    # row's result does not depend on its tile position -> identical bits
    idxs, cnts = ops.ball_query_multi([radius], [S], _t(xyz, dev), _t(new_xyz, dev), return_counts=True)
    np.testing.assert_array_equal(idxs[0].cpu().numpy(), idx)
    m.default_geometry = 32 if m.preferred_geometry == 2 else 0      # the tiled kernel (an explicit tile height)
    packed = m.grouped(_t(xyz, dev), ft, _t(new_xyz, dev), idxs[0], cnt=cnts[0]).cpu().numpy()
    if m.default_geometry == 0:
        assert np.array_equal(packed, got), f"packed vs dense: {np.abs(packed - got).max()}"
    else:
        _close(packed, want, "packed rows, tiled kernel")
    m.default_geometry = 0
    if m.preferred_geometry == 2:    # compiled shape: calls with counts run the register-resident chain (other summation order)
        reg = m.grouped(_t(xyz, dev), ft, _t(new_xyz, dev), idxs[0], cnt=cnts[0]).cpu().numpy()
        _close(reg, want, "register-resident chain (geometry 2)")


REG_BF16 = [
    # (B, N, M, S, C, mlp, radius, feat dtype) — every compiled shape of the register-resident bf16 chain (geometry 2)
    (2, 3000, 700, 32, 1, [16, 16, 32], 0.08, "f32"),       # SA1 narrow (KITTI: one f32 channel, strided view)
    (8, 8192, 4096, 32, 1, [16, 16, 32], 0.05, "f32"),      # ... many items per workgroup
    (2, 3000, 700, 64, 1, [32, 32, 64], 0.15, "f32"),       # SA1 wide, nsample 64
    (2, 3000, 500, 32, 4, [16, 16, 32], 0.1, "f32"),        # nuScenes SA1 (4 f32 channels, row stride 7)
    (1, 1024, 256, 32, 0, [64, 64, 128], 0.2, None),        # BASELINE configs[0] (no features)
    (2, 2000, 400, 32, 64, [64, 64, 128], 0.2, "bf16"),     # SA2
    (2, 2000, 400, 64, 64, [64, 96, 128], 0.35, "bf16"),
    (2, 1024, 300, 32, 128, [128, 128, 256], 0.3, "bf16"),  # SA3
    (2, 1024, 300, 32, 128, [128, 192, 256], 0.5, "bf16"),
    (2, 1024, 300, 32, 128, [128, 256, 256], 0.7, "bf16"),
    (2, 512, 256, 16, 256, [256, 256, 512], 0.25, "bf16"),  # cluster
    (2, 512, 256, 32, 256, [256, 512, 1024], 0.35, "bf16"),
    (1, 200, 19, 8, 1, [16, 16, 32], 0.5, "f32"),           # fewer rows than one tile
    (3, 2048, 512, 32, 128, [128, 128, 256], 0.9, "bf16"),  # full groups: a group = a whole tile, many tiles per workgroup
]


@pytest.mark.parametrize("B,N,M,S,C,mlp,r,fdt", REG_BF16)
def test_register_chain_bf16(orc, sad, dev, B, N, M, S, C, mlp, r, fdt):
    """geometry 2 of sad_mlp_chain_bf16 (csrc/mlp_bf16_reg.hip): one wave carries a 32-row tile through the chain in
    registers.  Ragged groups, groups that straddle half-tiles and tiles (atomic max merge), rows past the end;
    with the chain's own row-packing scan and with a table made by rowscan_multi.  SPEC.md §14 tolerance."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(N + M + S + C + sum(mlp))
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, Cn = _t(xyz, dev), _t(new_xyz, dev)
    feat, F = None, None
    if C:
        if fdt == "f32":     # a strided view of the point rows, as the detector's first stage reads them
            pts = rng.uniform(0, 1, (B, N, 3 + C)).astype(np.float32)
            pts[:, :, 3:] = orc.bf16_round(pts[:, :, 3:])
            feat = np.ascontiguousarray(pts[:, :, 3:])
            F = _t(pts, dev)[:, :, 3:]
        else:
            feat = orc.bf16_round(rng.normal(size=(B, N, C)).astype(np.float32))
            F = _t(feat, dev).bfloat16()
    idxs, cnts = ops.ball_query_multi((r,), (S,), X, Cn, return_counts=True)
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    want = orc.sa_group_mlp_max_bf16(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
    net = ops.PackedMLPBf16(layers, True, dev)
    assert net.preferred_geometry == 2
    net.default_geometry = 2
    got = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
    rows = int(cnts[0].clamp(min=1).sum().item())
    _close(got, want, f"register chain bf16 {[C + 3] + mlp} S={S} ({rows} packed rows)")
    ws = ops.rowscan_multi(idxs, cnts, N)[0]
    again = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0], ws=ws).cpu().numpy()
    assert np.array_equal(again, got), "prescanned table: different bits"


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_register_chain_bf16_random_group_sizes(orc, sad, dev, seed):
    """Arbitrary group sizes (not from a ball query): every mix of 1 .. S rows per group, so tiles with 1, 8, 9, 16, 17 and 32
    groups, groups that span two or three tiles, a last tile with a single live row — the staged pooling's slot / block-width
    choices (CB = 128 / 64 / 32) and its atomic merge of tile-crossing groups all occur.  SPEC.md §14 tolerance vs the oracle."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(1000 + seed)
    B, N, C = 2, 900, 64
    S = int(rng.choice([8, 16, 32, 64]))
    M = int(rng.integers(40, 400))
    mlp = [[64, 64, 128], [64, 96, 128]][seed % 2]
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = orc.bf16_round(rng.normal(size=(B, N, C)).astype(np.float32))
    new_xyz = rng.uniform(0, 1, (B, M, 3)).astype(np.float32)
    idx = rng.integers(0, N, (B, M, S)).astype(np.int32)
    mode = seed % 3
    if mode == 0:
        cnt = rng.integers(1, S + 1, (B, M))
    elif mode == 1:
        cnt = rng.choice([1, 1, 1, 2, S], size=(B, M))                   # many single-row groups (32 groups per tile) and full ones
    else:
        cnt = np.where(rng.random((B, M)) < 0.5, S, rng.integers(1, 4, (B, M)))
    cnt = cnt.astype(np.int32)
    for b in range(B):
        for m in range(M):
            idx[b, m, cnt[b, m]:] = idx[b, m, 0]                          # ball-query style padding
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    want = orc.sa_group_mlp_max_bf16(xyz, feat, new_xyz, idx, layers)
    net = ops.PackedMLPBf16(layers, True, dev)
    net.default_geometry = 2
    got = net.grouped(_t(xyz, dev), _t(feat, dev).bfloat16(), _t(new_xyz, dev), _t(idx, dev), cnt=_t(cnt, dev)).cpu().numpy()
    _close(got, want, f"random groups seed {seed}: S={S} M={M} mode={mode} rows={int(cnt.sum())}")


@pytest.mark.parametrize("seed,C,mlp", [(11, 64, [64, 64, 128]), (12, 64, [64, 96, 128]), (13, 256, [256, 256, 512])])
def test_register_chain_bf16_unaligned_output_slice(sad, dev, seed, C, mlp):
    """The register-resident chain writing into columns [5, 5 + C_out) of a buffer whose row stride is not a multiple of four: no 16-byte
    store is possible, whole groups leave through the element-wise path and groups that cross a tile through the per-channel atomic max with its
    own bounds (csrc/mlp_bf16_reg.hip, flush of a block).  Same arithmetic as the aligned call: identical bits; nothing outside the slice is written."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(2000 + seed)
    B, N = 2, 700
    S = int(rng.choice([16, 32]))
    M = int(rng.integers(60, 200))
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = rng.uniform(0, 1, (B, M, 3)).astype(np.float32)
    idx = rng.integers(0, N, (B, M, S)).astype(np.int32)
    cnt = rng.choice([1, 2, 3, S // 2, S], size=(B, M)).astype(np.int32)       # tiles of many groups and of few, groups across tiles
    for b in range(B):
        for m in range(M):
            idx[b, m, cnt[b, m]:] = idx[b, m, 0]
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    net = ops.PackedMLPBf16(layers, True, dev)
    assert net.preferred_geometry == 2
    net.default_geometry = 2
    X, F, Cn, I, K = _t(xyz, dev), _t(feat, dev).bfloat16(), _t(new_xyz, dev), _t(idx, dev), _t(cnt, dev)
    want = net.grouped(X, F, Cn, I, cnt=K)
    cout, off = mlp[-1], 5
    buf = torch.full((B, M, cout + 13), -7.0, device=dev, dtype=torch.float32)
    buf[:, :, off:off + cout] = 0.0                 # (the slice starts at zero: groups that cross a tile are merged with an atomic max)
    assert buf.stride(1) % 4 != 0
    net.grouped(X, F, Cn, I, out=buf, col_off=off, cnt=K)
    torch.cuda.synchronize()
    assert torch.equal(buf[:, :, off:off + cout], want), "unaligned slice: different bits from the aligned call"
    assert bool((buf[:, :, :off] == -7).all()) and bool((buf[:, :, off + cout:] == -7).all()), "written outside the slice"


def test_register_chain_bf16_three_chain_dispatch(orc, sad, dev):
    """The three SA3 branches as ONE register-resident dispatch (sad_mlp_chain_multi_bf16) with prescanned tables:
    chains of different shapes follow each other in a workgroup's item list and the weight ring carries over."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(78)
    B, N, M, C = 3, 1024, 384, 128
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = orc.bf16_round(rng.normal(size=(B, N, C)).astype(np.float32))
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, F, Cn = _t(xyz, dev), _t(feat, dev).bfloat16(), _t(new_xyz, dev)
    radii, ns = (0.2, 0.35, 0.6), (32, 32, 32)
    mlps = ([128, 128, 256], [128, 192, 256], [128, 256, 256])
    idxs, cnts = ops.ball_query_multi(radii, ns, X, Cn, return_counts=True)
    wss = ops.rowscan_multi(idxs, cnts, N)
    out = torch.zeros((B, M, 768), device=dev)
    calls, wants = [], []
    for bi, mlp in enumerate(mlps):
        layers = synth.make_mlp_weights([C + 3] + mlp, rng)
        net = ops.PackedMLPBf16(layers, True, dev)
        calls.append((net, X, F, Cn, idxs[bi], out, 256 * bi, cnts[bi], wss[bi]))
        wants.append(orc.sa_group_mlp_max_bf16(xyz, feat, new_xyz, idxs[bi].cpu().numpy(), layers))
    ops.grouped_multi(calls)
    got = out.cpu().numpy()
    for bi in range(3):
        _close(got[:, :, 256 * bi:256 * (bi + 1)], wants[bi], f"merged dispatch, branch {bi}")
    out2 = torch.zeros((B, M, 768), device=dev)       # the same tables again (nothing in them is consumed)
    ops.grouped_multi([c[:5] + (out2,) + c[6:] for c in calls])
    assert torch.equal(out, out2)


def test_nuscenes_stage_bf16(orc, sad, dev):
    """configs[4] shape: one SA1 branch on a 65 536-point nuScenes-shaped scene, bf16 MLP; index
    operators stay bit-exact (checked in test_gpu_ops), the pooled features meet the §14 tolerance."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(65536)
    N, M, S, C = 65536, 2048, 32, 8
    xyz = np.empty((1, N, 3), np.float32)
    xyz[0, :, 0] = rng.uniform(-51.2, 51.2, N)
    xyz[0, :, 1] = rng.uniform(-51.2, 51.2, N)
    xyz[0, :, 2] = rng.normal(-1.6, 0.2, N)
    feat = orc.bf16_round(rng.normal(size=(1, N, C)).astype(np.float32))
    layers = synth.make_mlp_weights([C + 3, 32, 32, 64], rng)
    xt = _t(xyz, dev)
    fidx = ops.fps(xt, M)
    new_xyz = ops.gather_xyz(xt, fidx)
    idx = ops.ball_query(0.8, S, xt, new_xyz)
    got = ops.PackedMLPBf16(layers, True, dev).grouped(xt, _t(feat, dev).bfloat16(), new_xyz, idx).cpu().numpy()
    want = orc.sa_group_mlp_max_bf16(xyz, feat, new_xyz.cpu().numpy(), idx.cpu().numpy(), layers)
    _close(got, want, "nuScenes-shaped SA1 branch")


def _stage_oracle(orc, xyz, feat, new_xyz, idxs, weights, name, agg):
    """One SA / cluster stage of SPEC §14 from given inputs: branches -> concat -> aggregation."""
    cat = np.concatenate([orc.sa_group_mlp_max_bf16(xyz, feat, new_xyz, idx, weights[f"{name}.b{i}"])
                          for i, idx in enumerate(idxs)], axis=2)
    B, M, C = cat.shape
    out = orc.mlp_rows_bf16(cat.reshape(B * M, C), weights[f"{name}.agg"]).reshape(B, M, -1)
    return cat, out


@pytest.mark.parametrize("cfg_name,batch", [("TINY", 2), ("KITTI", 1), ("NUSCENES", 1)])
def test_detector_bf16_stagewise(orc, sad, dev, cfg_name, batch):
    """The whole path with dtype="bf16", each stage checked against the §14 oracle fed with the
    GPU's own upstream tensors (teacher forcing: a bf16 rounding flip must not be allowed to move
    a later index decision and then be counted as an error).  Index operators are checked
    bit-exactly on those same inputs."""
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    cfg = getattr(config, cfg_name)
    w = synth.make_weights(cfg, 0)
    pts = {"TINY": lambda: synth.make_tiny_batch(0, batch, cfg.n_points), "KITTI": lambda: synth.make_batch(0, batch),
           "NUSCENES": lambda: synth.make_nuscenes_batch(0, batch)}[cfg_name]()   # configs[4]: 65 536 points
    det = SADDetector(cfg, w, dev, dtype="bf16")
    tr = {}
    boxes = det(_t(pts, dev), tr)
    torch.cuda.synchronize()
    xyz = np.ascontiguousarray(pts[:, :, :3])
    feat = np.ascontiguousarray(pts[:, :, 3:])
    for si, st in enumerate(cfg.stages):
        name = f"sa{si + 1}"
        t = tr[name]
        new_xyz = t["new_xyz"].cpu().numpy()
        np.testing.assert_array_equal(new_xyz, orc.gather_xyz(xyz, orc.fps(xyz, st.npoint)))
        idxs = [i.cpu().numpy() for i in t["ball_idx"]]
        for r, s, got in zip(st.radii, st.nsamples, idxs):
            np.testing.assert_array_equal(got, orc.ball_query(r, s, xyz, new_xyz))
        _, want = _stage_oracle(orc, xyz, orc.bf16_round(feat), new_xyz, idxs, w, name, st.agg)
        assert t["out"].dtype == torch.bfloat16
        got = t["out"].float().cpu().numpy()
        _close(got, orc.bf16_round(want), f"{cfg_name} {name} bf16 features")
        xyz, feat = new_xyz, got
    c = tr["cluster"]
    K = cfg.n_cand
    want_c = orc.mlp_rows_bf16(feat[:, :K].reshape(-1, feat.shape[2]), w["cand"],
                               relu_mask=(1 << (len(w["cand"]) - 1)) - 1).reshape(batch, K, -1)
    gc = c["c"].cpu().numpy()
    _close(gc, want_c, "candidate MLP")
    cand, rad = orc.candidates(xyz, gc, cfg.shift_max, cfg.r_min, cfg.r_max, cfg.anchor_car)
    np.testing.assert_array_equal(c["cand"].cpu().numpy(), cand)
    np.testing.assert_array_equal(c["radius"].cpu().numpy(), rad)
    idxs = [i.cpu().numpy() for i in c["ball_idx"]]
    for sc, s, got in zip(cfg.cluster_scales, cfg.cluster_nsamples, idxs):
        np.testing.assert_array_equal(got, orc.ball_query((np.float32(sc) * rad).astype(np.float32), s, xyz, cand))
    want_cat, want_cf = _stage_oracle(orc, xyz, feat, cand, idxs, w, "cluster", True)
    _close(c["cat"].cpu().numpy(), want_cat, "cluster pooled")
    gcf = c["cfeat"].float().cpu().numpy()
    _close(gcf, want_cf, "cluster features")
    want_h = orc.mlp_rows_bf16(gcf.reshape(batch * K, -1), w["head"],
                               relu_mask=(1 << (len(w["head"]) - 1)) - 1).reshape(batch, K, -1)
    gh = c["head"].cpu().numpy()
    _close(gh, want_h, "head")
    want_b = orc.decode_boxes(cand, gh, [v for a in cfg.anchors for v in a])
    b = boxes.cpu().numpy()
    np.testing.assert_array_equal(b[..., 8], want_b[..., 8])
    assert np.abs(b - want_b).max() <= 1e-4 * (1 + np.abs(want_b).max())


def test_golden_extensions_gpu(sad, dev):
    """The committed vectors of tests/golden/extensions.npz (SPEC §13-§15), without the live oracle."""
    import os
    from conftest import GOLDEN
    from sad_amd import ops
    g = np.load(os.path.join(GOLDEN, "extensions.npz"))
    xyz, feat = g["ffps_xyz"], g["ffps_feat"]
    np.testing.assert_array_equal(ops.ffps(_t(xyz, dev), _t(feat, dev), 64, 1.0).cpu().numpy(), g["ffps_idx_w1"])
    np.testing.assert_array_equal(ops.ffps(_t(xyz, dev), _t(feat, dev), 64, 0.0).cpu().numpy(), g["ffps_idx_w0"])
    layers = [(g[f"bf16_w{i}"], g[f"bf16_b{i}"]) for i in range(2)]
    new_xyz = np.ascontiguousarray(xyz[:, :50])
    idx = ops.ball_query(0.2, 16, _t(xyz, dev), _t(new_xyz, dev))
    np.testing.assert_array_equal(idx.cpu().numpy(), g["bf16_idx"])
    got = ops.PackedMLPBf16(layers, True, dev).grouped(_t(xyz, dev), _t(feat, dev), _t(new_xyz, dev), idx).cpu().numpy()
    _close(got, g["bf16_pooled"], "golden bf16 pooled")
    rows = np.concatenate([xyz[0, :32], feat[0, :32]], 1)
    _close(ops.PackedMLPBf16(layers, False, dev).rows(_t(rows, dev)).cpu().numpy(), g["bf16_rows"], "golden bf16 rows")
    keep, order, count = ops.nms_bev(_t(g["nms_boxes"], dev), 0.1, 0.2)
    np.testing.assert_array_equal(keep.cpu().numpy(), g["nms_keep"])
    np.testing.assert_array_equal(order.cpu().numpy(), g["nms_order"])
    np.testing.assert_array_equal(count.cpu().numpy(), g["nms_count"])


def test_bf16_rows_per_tile_geometries_agree(orc, sad, dev):
    """The rows-per-tile geometries the bf16 autotuner tries (32..256) give identical bits (a row's
    result does not depend on its tile), packed and dense; geometries that do not fit LDS are refused."""
    import ctypes
    import torch
    from sad_amd import ops, synth, _lib
    rng = np.random.default_rng(256)
    B, N, M, S, C = 2, 2000, 150, 32, 16
    xyz = rng.random((B, N, 3), dtype=np.float32)
    feat = orc.bf16_round(rng.normal(size=(B, N, C)).astype(np.float32))
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    idxs, cnts = ops.ball_query_multi([0.12], [S], _t(xyz, dev), _t(new_xyz, dev), return_counts=True)
    layers = synth.make_mlp_weights([C + 3, 32, 64], rng)
    net = ops.PackedMLPBf16(layers, True, dev)
    ref = None
    for geom in (0, 32, 64, 128, 256):
        for cnt in (None, cnts[0]):
            a, out, keep = net._grouped_args(_t(xyz, dev), _t(feat, dev).bfloat16(), _t(new_xyz, dev), idxs[0], None, 0, cnt)
            a.geometry = geom
            rc = _lib.lib().sad_mlp_chain_bf16(ctypes.byref(a), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, _lib.lib().sad_last_error()
            got = out.cpu().numpy()
            if ref is None:
                ref = got
                _close(got, orc.sa_group_mlp_max_bf16(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers), "geometry 0")
            assert np.array_equal(got, ref), f"geometry {geom} cnt={'yes' if cnt is not None else 'no'}"
    a, out, keep = net._grouped_args(_t(xyz, dev), _t(feat, dev).bfloat16(), _t(new_xyz, dev), idxs[0], None, 0, None)
    a.geometry = 100
    assert _lib.lib().sad_mlp_chain_bf16(ctypes.byref(a), torch.cuda.current_stream().cuda_stream) == -1


def test_detector_bf16_overlapped_path_with_poisoned_pooling_buffers(sad, dev):
    """The overlapped path (submit(): scans on the sampling streams) never zero-fills its pooling buffers — the row-packing
    scans prepare the groups that are combined with an atomic max.  With NaN in those buffers it must give the boxes of the
    serial path (same kernels, zero-filled buffers) bit for bit."""
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    cfg = config.KITTI
    w = synth.make_weights(cfg, 0)
    P = _t(synth.make_batch(0, 4), dev)
    ref = SADDetector(cfg, w, dev, overlap_fps=False, dtype="bf16")(P).clone()
    det = SADDetector(cfg, w, dev, overlap_fps=True, dtype="bf16")
    det.poison_buffers = True
    outs = [det.submit(P)[0] for _ in range(3)]
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref), "overlapped bf16 path with uninitialised pooling buffers differs from the serial path"



# ---- split pooling (include/sad_amd.h sad_mlp_bf16_args.cont, ABI 4) ------------------------------------------------------------------
# A split-pooled chain stores bf16 rows with plain stores: a group's rows in the tile where its packed rows begin pool into out[g], its
# rows in a later 32-row tile t into row t of the continuation buffer.  The same kernel arithmetic as the f32 / atomic-max form, and
# rounding to bf16 is monotone, so   max(out[g], cont[t] ...) == bf16(f32-pooled[g])   BIT FOR BIT, and a layer that reads the pooled
# rows (rows(pool=...)) must give the bits it gives on the f32 buffer.
def _gstart_off(ng, S):
    o = 4 + (ng + 1) + (ng * S // 32 + 2) + (ng // 1024 + 2) + 2 * ng * S
    return (o + 3) & ~3


SPLIT = [
    # (B, N, M, S, C, mlp, fill)    fill: share of the nsample a group gets on average (1.0: every group full -> 64-row groups span three tiles)
    (2, 2048, 512, 32, 1, [16, 16, 32], 0.15),
    (2, 2048, 512, 64, 1, [32, 32, 64], 0.1),
    (2, 1024, 256, 64, 64, [64, 96, 128], 1.0),
    (2, 1024, 256, 32, 64, [64, 64, 128], 0.3),
    (1, 512, 128, 32, 128, [128, 192, 256], 0.5),
    (1, 512, 100, 16, 256, [256, 256, 512], 0.6),
    (1, 512, 64, 32, 256, [256, 512, 1024], 1.0),
    (1, 512, 37, 64, 0, [64, 64, 128], 0.05),
]


def _split_case(sad, dev, B, N, M, S, C, mlp, fill, seed=0):
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(N + M + S + C + seed)
    xyz = rng.random((B, N, 3), dtype=np.float32)
    new_xyz = xyz[:, :M].copy() if M <= N else rng.random((B, M, 3), dtype=np.float32)
    idx = rng.integers(0, N, size=(B, M, S)).astype(np.int32)
    cnt = np.clip(rng.poisson(fill * S, size=(B, M)), 1, S).astype(np.int32) if fill < 1.0 else np.full((B, M), S, np.int32)
    if fill >= 1.0:
        cnt[:, ::5] = 13                                 # (full groups off the tile grid: 64-row groups then span three tiles)
    cnt[0, 0] = S
    cnt[-1, -1] = 1
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    m = ops.PackedMLPBf16(layers, True, dev)
    feat = _t(rng.normal(size=(B, N, C)).astype(np.float32), dev).bfloat16() if C else None
    return m, _t(xyz, dev), feat, _t(new_xyz, dev), _t(idx, dev), _t(cnt, dev), cnt


@pytest.mark.parametrize("B,N,M,S,C,mlp,fill", SPLIT)
def test_split_pooled_rows_are_the_f32_pooled_rows_rounded(sad, dev, B, N, M, S, C, mlp, fill):
    import torch
    from sad_amd import ops
    m, xyz, feat, new_xyz, idx, cnt, cnt_h = _split_case(sad, dev, B, N, M, S, C, mlp, fill)
    assert m.preferred_geometry == 2
    co = mlp[-1]
    ld, off = co + 24, 16                                # a slice of a wider buffer
    want = torch.zeros((B, M, ld), device=dev)
    m.grouped(xyz, feat, new_xyz, idx, out=want, col_off=off, cnt=cnt)
    want = want[:, :, off:off + co].bfloat16().view(torch.int16).cpu().numpy().reshape(B * M, co)
    out = torch.full((B, M, ld), -3.0, device=dev, dtype=torch.bfloat16)
    cont = ops.cont_buffer(B, M, S, co, dev)
    cont.fill_(0x7F)                                     # (poison: whatever is read must have been written)
    m.grouped(xyz, feat, new_xyz, idx, out=out, col_off=off, cnt=cnt, cont=cont)        # (scans for itself: prescanned = 0)
    # the same through an explicit split scan
    out2 = torch.full((B, M, ld), -3.0, device=dev, dtype=torch.bfloat16)
    cont2 = ops.cont_buffer(B, M, S, co, dev)
    cont2.fill_(0x7F)
    (ws2,) = ops.rowscan_multi([idx], [cnt], N, [(out2, off, co, cont2)])
    m.grouped(xyz, feat, new_xyz, idx, out=out2, col_off=off, cnt=cnt, ws=ws2, cont=cont2)
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(cont, cont2)
    o = out.view(torch.int16).cpu().numpy().reshape(B * M, ld)
    assert (out.float().cpu().numpy().reshape(B * M, ld)[:, :off] == -3).all() and (out.float().cpu().numpy().reshape(B * M, ld)[:, off + co:] == -3).all()
    main = o[:, off:off + co].astype(np.int32)           # non-negative bf16: ordered like their bit patterns
    ng = B * M
    tab = ws2.view(torch.int32).cpu().numpy()
    gs = tab[_gstart_off(ng, S):_gstart_off(ng, S) + ng + 1]
    c = cnt_h.reshape(-1)
    assert np.array_equal(np.diff(gs), c) and gs[0] == 0
    ct = cont2.view(torch.int16).cpu().numpy().reshape(-1, co).astype(np.int32)
    assert (ct[0] == 0).all()
    full = main.copy()
    t0, t1 = gs[:-1] >> 5, (gs[1:] - 1) >> 5
    n_cont = t1 - t0
    for k in (1, 2):
        sel = n_cont >= k
        full[sel] = np.maximum(full[sel], ct[t0[sel] + k])
    assert n_cont.max() <= 2 and (n_cont > 0).any()
    if fill >= 1.0 and S == 64:
        assert (n_cont == 2).any()
    assert np.array_equal(full.astype(np.int16), want), "split-pooled rows differ from the rounded f32-pooled rows"
    # straddling groups really are split (the main row alone is not the answer)
    assert not np.array_equal(main.astype(np.int16), want)


@pytest.mark.parametrize("n_chain", [1, 3, 4, 30])
def test_layer_reading_split_pooled_rows_equals_layer_on_f32_pooled_rows(sad, dev, n_chain):
    """sa2-like stage: three branches into one [B,M,384] buffer, aggregation 384 -> 128, f32 pooling against split pooling; then the
    same with full 64-row groups (second continuation rows) and a deep layer on the queued loop."""
    import torch
    from sad_amd import ops, synth
    big = n_chain == 30          # three branches on 40 002 groups: more row blocks than CUs -> the persistent form of the reading layer
    n_chain = 3 if big else n_chain
    specs = [(32, [64, 64, 128], 0.3), (32, [64, 64, 128], 0.6), (64, [64, 96, 128], 1.0), (16, [64, 64, 128], 0.5)][:n_chain]
    B, N, M, C = 2, 2048, (20001 if big else (641 if n_chain == 4 else 640)), 64       # (four branches: two dispatches of the chain kernel, a ragged last row block)
    rng = np.random.default_rng(5 + n_chain)
    cat_c = sum(s[1][-1] for s in specs)
    agg = ops.PackedMLPBf16(synth.make_mlp_weights([cat_c, 128], rng), False, dev)
    cat32 = torch.zeros((B, M, cat_c), device=dev)
    cat16 = torch.full((B, M, cat_c), 9.0, device=dev, dtype=torch.bfloat16)
    calls32, calls16, outs, idxs, cnts, conts, off = [], [], [], [], [], [], 0
    for i, (S, mlp, fill) in enumerate(specs):
        m, xyz, feat, new_xyz, idx, cnt, _ = _split_case(sad, dev, B, N, M, S, C, mlp, fill, seed=i)
        cont = ops.cont_buffer(B, M, S, mlp[-1], dev)
        cont.fill_(0x7F)
        calls32.append((m, xyz, feat, new_xyz, idx, cat32, off, cnt))
        calls16.append([m, xyz, feat, new_xyz, idx, cat16, off, cnt, None, cont])
        outs.append((cat16, off, mlp[-1], cont))
        idxs.append(idx); cnts.append(cnt); conts.append(cont)
        off += mlp[-1]
    for c in calls32:
        c[0].grouped(*c[1:5], out=c[5], col_off=c[6], cnt=c[7])
    wss = ops.rowscan_multi(idxs, cnts, N, outs)
    for c, w in zip(calls16, wss):
        c[8] = w
    ops.grouped_multi([tuple(c) for c in calls16]) if n_chain > 1 else calls16[0][0].grouped(*calls16[0][1:5], out=cat16, col_off=0, cnt=cnts[0], ws=wss[0], cont=conts[0])
    want = agg.rows(cat32, out_dtype=torch.bfloat16)
    pool = [(w, k, s[0], s[1][-1]) for w, k, s in zip(wss, conts, specs)]
    got = agg.rows(cat16, out_dtype=torch.bfloat16, pool=pool)
    torch.cuda.synchronize()
    assert torch.equal(got, want), f"max diff {(got.float() - want.float()).abs().max().item()}"
    got32 = agg.rows(cat16, pool=pool)
    assert torch.equal(got32, agg.rows(cat32))
    # refused where it cannot run: a two-layer chain, f32 rows
    two = ops.PackedMLPBf16(synth.make_mlp_weights([cat_c, 64, 32], rng), False, dev)
    with pytest.raises(RuntimeError):
        two.rows(cat16, pool=pool)


@pytest.mark.parametrize("cfg_name", ["TINY", "KITTI", "NUSCENES"])
def test_detector_bf16_split_pooling_changes_no_bit(sad, dev, cfg_name):
    import torch
    from sad_amd import config, ops, synth
    from sad_amd.detector import SADDetector
    cfg = getattr(config, cfg_name)
    w = synth.make_weights(cfg, 0)
    B = {"TINY": 3, "KITTI": 4, "NUSCENES": 2}[cfg_name]
    make = {"TINY": synth.make_tiny_batch, "KITTI": synth.make_batch, "NUSCENES": synth.make_nuscenes_batch}[cfg_name]
    pts = _t(make(21, B, cfg.n_points), dev)
    old = ops.SPLIT_POOL
    try:
        ops.SPLIT_POOL = False
        ref_det = SADDetector(cfg, w, dev, dtype="bf16")
        ref_det.use_plans = False
        ref, ev = ref_det.submit(pts)
        ev.synchronize()
        ops.SPLIT_POOL = True
        det = SADDetector(cfg, w, dev, dtype="bf16", streams=(ref_det._sides, ref_det._mains))
        for i in range(det._plan_ring + 3):              # recorded and replayed steps
            out, ev = det.submit(pts)
        ev.synchronize()
        assert det.plan_refused is None and det.plan_replays >= 3
    finally:
        ops.SPLIT_POOL = old
    assert torch.equal(out, ref)
    # the split path really ran: every stage with an aggregation layer and the cluster layer
    n_in = [cfg.n_points] + [s.npoint for s in cfg.stages[:-1]]
    assert all(m.can_split(B, n, m.stage.npoint, feat_dtype=torch.bfloat16 if i else torch.float32) for i, (m, n) in enumerate(zip(det.stages, n_in)))


# ---- the row-streaming layer's second form (csrc/mlp_bf16_rows.hip: a tiled GEMM fed by LDS-DMA) --------------------------------------
# Serves bf16 rows whose K is a multiple of 64; the same products in the same k order per accumulator as the first form, so the two must
# agree BIT FOR BIT (sad_set_option("mlp_rows_form", 1) forces the first form).
ROWS2 = [
    (1000, [128, 64]),          # two chunks, two channel tiles (NTW = 1), ragged last row block
    (8192, [1536, 512]),        # cluster.agg: 24 chunks, four channel blocks
    (16384, [768, 256]),        # sa3.agg
    (130, [384, 128]),          # two row blocks, the second nearly empty
    (4099, [64, 96]),           # ONE chunk; three channel tiles: a ragged channel block
    (257, [192, 40]),           # cout not a multiple of 32
    (70000, [128, 64]),         # more row blocks than CUs: the persistent third form (two chunks per item, ragged last block)
    (200001, [384, 128]),       # third form, six items per workgroup, four channel tiles
    (40000, [1536, 512]),       # third form, four channel blocks (313 row blocks x 4 on 256 workgroups)
]


@pytest.mark.parametrize("rows,dims", ROWS2)
def test_rows_bf16_second_form_equals_first_form(orc, sad, dev, rows, dims):
    import torch
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(rows + sum(dims))
    layers = synth.make_mlp_weights(dims, rng)
    x = _t(rng.normal(size=(rows, dims[0])).astype(np.float32), dev).bfloat16()
    mlp = ops.PackedMLPBf16(layers, False, dev)
    try:
        _lib.set_option("mlp_rows_form", 1)
        want32 = mlp.rows(x)
        want16 = mlp.rows(x, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("mlp_rows_form", 0)
    got32 = mlp.rows(x)
    got16 = mlp.rows(x, out_dtype=torch.bfloat16)
    buf = torch.full((rows, dims[-1] + 24), -7.0, device=dev, dtype=torch.bfloat16)
    mlp.rows(x, out=buf, col_off=8)
    torch.cuda.synchronize()
    assert torch.equal(got32, want32), f"f32 out: max diff {(got32 - want32).abs().max().item()}"
    assert torch.equal(got16, want16)
    assert torch.equal(buf[:, 8:8 + dims[-1]], want16) and (buf[:, :8] == -7).all() and (buf[:, 8 + dims[-1]:] == -7).all()
    _close(got32.cpu().numpy(), orc.mlp_rows_bf16(x.float().cpu().numpy(), layers), f"rows2 {dims} x {rows}")


def test_bf16_pipelined_steps_soak(sad, dev):
    """The LDS-DMA kernels (chain weight ring, row-streaming layer) and split pooling under the pipelined submit path: rotating KITTI-shaped
    batches, several steps in flight, every result bit-equal to the eager single-step result of its batch (a DMA read too early would show
    as a rare wrong tile).  SAD_STRESS=20: a longer soak of the same check."""
    import os
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    stress = int(os.environ.get("SAD_STRESS", "1"))
    cfg = config.KITTI
    w = synth.make_weights(cfg, 0)
    batches = [_t(synth.make_batch(100 + 8 * k, 8, cfg.n_points), dev) for k in range(4)]
    ref_det = SADDetector(cfg, w, dev, dtype="bf16")
    ref_det.use_plans = False
    want = []
    for b in batches:
        out, ev = ref_det.submit(b)
        ev.synchronize()
        want.append(out.clone())
        torch.cuda.current_stream().synchronize()        # (the copy is on the null stream: done before the next step reuses `out`'s block)
    det = SADDetector(cfg, w, dev, dtype="bf16", streams=(ref_det._sides, ref_det._mains))
    pending, bad = [], 0
    for i in range(60 * stress):
        k = (i * 3 + i // 7) % 4
        out, ev = det.submit(batches[k])
        pending.append((out, ev, k, i))
        if len(pending) > 6:
            o, e, kk, ii = pending.pop(0)
            e.synchronize()
            bad += int(not torch.equal(o, want[kk]))
    for o, e, kk, ii in pending:
        e.synchronize()
        bad += int(not torch.equal(o, want[kk]))
    assert bad == 0, f"{bad} of {60 * stress} pipelined steps differ from the eager result of their batch"
    assert det.plan_refused is None
