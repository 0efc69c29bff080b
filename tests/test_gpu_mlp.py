"""GPU parity of the fused group -> MLP -> max kernel, sa_module and the detector (-m gpu)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _close(got, want, what):
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    rel = diff / (1.0 + np.abs(want))
    print(f"[parity] {what}: max|diff|={diff.max():.3e} bit-exact={np.array_equal(got, want)}")
    assert rel.max() <= TOL, f"{what}: max rel diff {rel.max():.3e} > {TOL}"


PLAIN_CASES = [
    # (rows, dims, relu_mask or None)
    (37, [7, 12, 5], None),
    (1000, [128, 64], None),
    (300, [384, 128], None),
    (257, [768, 256], None),
    (130, [1536, 512], None),           # k-chunked staging
    (256, [512, 256, 256, 10], 0b011),  # head: last layer linear
    (100, [256, 128, 6], 0b01),         # candidate MLP
    (64, [33, 40, 50, 60, 70], None),   # odd widths, 4 layers
]


@pytest.mark.parametrize("rows,dims,mask", PLAIN_CASES)
def test_mlp_rows_parity(orc, sad, dev, rows, dims, mask):
    from sad_amd import ops, synth
    rng = np.random.default_rng(sum(dims) + rows)
    layers = synth.make_mlp_weights(dims, rng)
    x = rng.normal(size=(rows, dims[0])).astype(np.float32)
    mlp = ops.PackedMLP(layers, False, dev, relu_mask=mask)
    got = mlp.rows(_t(x, dev)).cpu().numpy()
    want = orc.mlp_rows(x, layers, relu_mask=mask)
    _close(got, want, f"mlp_rows {dims}")
    # writing into a slice of a wider buffer leaves the rest untouched
    import torch
    buf = torch.full((rows, dims[-1] + 12), -7.0, device=dev)
    mlp.rows(_t(x, dev), out=buf, col_off=8)
    b = buf.cpu().numpy()
    _close(b[:, 8:8 + dims[-1]], want, "slice")
    assert (b[:, :8] == -7).all() and (b[:, 8 + dims[-1]:] == -7).all()


GROUPED_CASES = [
    # (B, N, M, S, C, mlp, radius)
    (1, 1024, 256, 32, 0, [64, 64, 128], 0.2),        # BASELINE configs[0]
    (2, 2048, 512, 32, 1, [16, 16, 32], 0.15),        # SA1 branch shape (C=1, strided feature view)
    (2, 2048, 512, 64, 1, [32, 32, 64], 0.3),         # nsample 64: pooling across two row tiles
    (2, 1024, 256, 32, 64, [64, 64, 128], 0.25),      # SA2
    (2, 1024, 256, 64, 64, [64, 96, 128], 0.4),
    (2, 512, 128, 32, 128, [128, 192, 256], 0.5),     # SA3
    (2, 256, 64, 16, 256, [256, 256, 512], 0.6),      # cluster branch 0 (nsample 16)
    (1, 256, 64, 32, 256, [256, 512, 1024], 0.8),     # cluster branch 1 (8 waves)
    (1, 300, 37, 24, 5, [20, 30], 0.4),               # odd everything: nsample 24 padded to 32
    (1, 200, 19, 8, 3, [8], 0.5),
]


@pytest.mark.parametrize("B,N,M,S,C,mlp,r", GROUPED_CASES)
def test_grouped_mlp_parity(orc, sad, dev, B, N, M, S, C, mlp, r):
    from sad_amd import ops, synth
    rng = np.random.default_rng(N + M + S + C)
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    fidx = orc.fps(xyz, M)
    new_xyz = orc.gather_xyz(xyz, fidx)
    idx = orc.ball_query(r, S, xyz, new_xyz)
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    X = _t(xyz, dev)
    if C == 1:   # like the detector: features are a strided view into the [B,N,4] point tensor
        pts = np.concatenate([xyz, rng.uniform(0, 1, (B, N, 1)).astype(np.float32)], -1)
        feat = np.ascontiguousarray(pts[:, :, 3:])
        F = _t(pts, dev)[:, :, 3:]
    elif C:
        feat = rng.normal(size=(B, N, C)).astype(np.float32)
        F = _t(feat, dev)
    else:
        feat, F = None, None
    got = ops.PackedMLP(layers, True, dev).grouped(X, F, _t(new_xyz, dev), _t(idx, dev)).cpu().numpy()
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idx, layers)
    _close(got, want, f"grouped C={C} S={S} mlp={mlp}")


@pytest.mark.parametrize("S,C,mlp", [(32, 16, [32, 32, 64]), (64, 64, [64, 96, 128]), (24, 5, [20, 30]), (16, 256, [256, 512])])
def test_grouped_mlp_arbitrary_indices(orc, sad, dev, S, C, mlp):
    """idx need not come from ball_query: random neighbours (every group full -> several passes per
    workgroup, groups straddling row tiles -> atomic-max combine), groups with random amounts of
    trailing padding, duplicates in the MIDDLE of a group (must still be computed) and the
    no-dedupe switch all give the oracle's pooled features."""
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(S * 7 + C)
    B, N, M = 2, 700, 300
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = rng.uniform(0, 1, (B, M, 3)).astype(np.float32)
    idx = rng.integers(0, N, (B, M, S)).astype(np.int32)
    cnt = rng.integers(1, S + 1, (B, M))
    cnt[:, ::3] = S                                     # a third of the groups are full
    for b in range(B):
        for m in range(M):
            idx[b, m, cnt[b, m]:] = idx[b, m, 0]        # trailing padding with the first index
    idx[0, 5, 2] = idx[0, 5, 0]                         # duplicate in the middle, real rows after it
    idx[1, 7, :] = 3                                    # one point repeated S times
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idx, layers)
    mlp_gpu = ops.PackedMLP(layers, True, dev)
    args = (_t(xyz, dev), _t(feat, dev), _t(new_xyz, dev), _t(idx, dev))
    _close(mlp_gpu.grouped(*args).cpu().numpy(), want, f"arbitrary idx S={S}")
    for key, val in (("mlp_nodedup", 1), ("mlp_dedup_f", 1), ("mlp_dedup_f", 64)):
        _lib.set_option(key, val)
        try:
            got = mlp_gpu.grouped(*args).cpu().numpy()
        finally:
            _lib.set_option(key, 0)
        _close(got, want, f"{key}={val}")


@pytest.mark.parametrize("rw", [1, 2, 4])
def test_grouped_mlp_rows_per_wave_option(orc, sad, dev, rw):
    """Every row-tiles-per-wave variant of the kernel gives the same answer."""
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(77)
    xyz = rng.uniform(0, 1, (2, 1024, 3)).astype(np.float32)
    feat = rng.normal(size=(2, 1024, 16)).astype(np.float32)
    new_xyz = orc.gather_xyz(xyz, orc.fps(xyz, 200))
    idx = orc.ball_query(0.2, 32, xyz, new_xyz)
    layers = synth.make_mlp_weights([19, 32, 32, 64], rng)
    _lib.set_option("mlp_rw", rw)
    try:
        got = ops.PackedMLP(layers, True, dev).grouped(_t(xyz, dev), _t(feat, dev), _t(new_xyz, dev), _t(idx, dev)).cpu().numpy()
    finally:
        _lib.set_option("mlp_rw", 0)
    _close(got, orc.sa_group_mlp_max(xyz, feat, new_xyz, idx, layers), f"rw={rw}")


def test_golden_config0_on_gpu(sad, dev):
    """BASELINE.json configs[0] against the COMMITTED golden vectors (no oracle build involved)."""
    from sad_amd import config, ops, synth
    from sad_amd.sa_module import SAModule
    g = np.load(os.path.join(GOLDEN, "config0.npz"))
    st = config.CONFIG0_SA
    X = _t(g["xyz"], dev)
    fidx = ops.fps(X, st.npoint)
    np.testing.assert_array_equal(fidx.cpu().numpy(), g["fps_idx"])
    new_xyz = ops.gather_xyz(X, fidx)
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), g["new_xyz"])
    bidx = ops.ball_query(st.radii[0], st.nsamples[0], X, new_xyz)
    np.testing.assert_array_equal(bidx.cpu().numpy(), g["ball_idx"])
    np.testing.assert_array_equal(ops.knn_query(16, X, new_xyz).cpu().numpy(), g["knn_idx"])
    layers = synth.make_mlp_weights([3, 64, 64, 128], np.random.default_rng(7))
    mod = SAModule(0, st.npoint, st.radii[0], st.nsamples[0], [64, 64, 128], dev, weights={"b0": layers})
    nx, nf = mod(X, None)                       # drop-in surface: features [B,C',M]
    np.testing.assert_array_equal(nx.cpu().numpy(), g["new_xyz"])
    assert tuple(nf.shape) == (1, 128, 256)
    _close(nf.transpose(1, 2).cpu().numpy(), g["feat"], "config0 sa_module")


def test_sa_module_msg_dropin_surface(orc, sad, dev):
    """sa_module(xyz [B,N,3], features [B,C,N]) -> (new_xyz, new_features [B,C',M]) vs the oracle."""
    from sad_amd import config, synth
    from sad_amd.sa_module import SAModuleMSG
    cfg = config.TINY
    st = cfg.stages[1]
    rng = np.random.default_rng(5)
    w = {f"b{i}": synth.make_mlp_weights([64 + 3] + list(m), rng) for i, m in enumerate(st.mlps)}
    w["agg"] = synth.make_mlp_weights([sum(m[-1] for m in st.mlps), st.agg], rng)
    pts = synth.make_tiny_batch(50, 2, 512)
    xyz = np.ascontiguousarray(pts[:, :, :3])
    feat_cm = rng.normal(size=(2, 64, 512)).astype(np.float32)
    mod = SAModuleMSG(64, st, dev, w)
    nx, nf = mod(_t(xyz, dev), _t(feat_cm, dev))
    ow = {f"sa.b{i}": w[f"b{i}"] for i in range(3)}
    ow["sa.agg"] = w["agg"]
    tr = {}
    onx, onf = orc.sa_module(xyz, np.ascontiguousarray(feat_cm.transpose(0, 2, 1)), st, ow, "sa", tr)
    np.testing.assert_array_equal(nx.cpu().numpy(), onx)
    assert tuple(nf.shape) == (2, st.agg, st.npoint)
    _close(nf.transpose(1, 2).cpu().numpy(), onf, "sa_module MSG")


def test_sa_module_msg_four_radii(orc, sad, dev):
    """a 4-radius stage (SAD_MAX_RADII = 4) on the default, non-autotuned path — the row-packing
    prescan serves four chains; branches 0-2 have compiled register-resident shapes, branch 3 ([48, 80]) has none
    and runs the tiled kernel, which must get no prescanned table.  Bit-exact vs the oracle."""
    from sad_amd import config, ops
    from sad_amd import synth
    from sad_amd.sa_module import SAModuleMSG
    assert not ops.AUTOTUNE
    st = config.SAStage(256, (0.8, 1.6, 2.4, 3.2), (32, 32, 64, 16),
                        ((64, 64, 128), (64, 64, 128), (64, 96, 128), (48, 80)), 96)
    rng = np.random.default_rng(11)
    w = {f"b{i}": synth.make_mlp_weights([64 + 3] + list(m), rng) for i, m in enumerate(st.mlps)}
    w["agg"] = synth.make_mlp_weights([sum(m[-1] for m in st.mlps), st.agg], rng)
    pts = synth.make_tiny_batch(70, 2, 1024)
    xyz = np.ascontiguousarray(pts[:, :, :3])
    feat_cm = rng.normal(size=(2, 64, 1024)).astype(np.float32)
    mod = SAModuleMSG(64, st, dev, w)
    X = _t(xyz, dev)
    q = mod.query(X, X[:, :st.npoint].contiguous(), prescan=True)
    assert [t is not None for t in q[2]] == [True, True, True, False], "prescan tables: only for kernels that consume them"
    nx, nf = mod(X, _t(feat_cm, dev))
    ow = {f"sa.b{i}": w[f"b{i}"] for i in range(4)}
    ow["sa.agg"] = w["agg"]
    onx, onf = orc.sa_module(xyz, np.ascontiguousarray(feat_cm.transpose(0, 2, 1)), st, ow, "sa", {})
    np.testing.assert_array_equal(nx.cpu().numpy(), onx)
    assert np.array_equal(nf.transpose(1, 2).cpu().numpy(), onf), "4-radius SA module differs from the oracle"
    # four chains of one compiled shape family: the prescan really carries four tables
    st4 = config.SAStage(256, (0.8, 1.6, 2.4, 3.2), (32, 32, 64, 32),
                         ((64, 64, 128), (64, 64, 128), (64, 96, 128), (64, 64, 128)), 0)
    w4 = {f"b{i}": synth.make_mlp_weights([64 + 3] + list(m), rng) for i, m in enumerate(st4.mlps)}
    mod4 = SAModuleMSG(64, st4, dev, w4)
    q4 = mod4.query(X, X[:, :st4.npoint].contiguous(), prescan=True)
    assert all(t is not None for t in q4[2])
    nx4, nf4 = mod4(X, _t(feat_cm, dev))
    onx4, onf4 = orc.sa_module(xyz, np.ascontiguousarray(feat_cm.transpose(0, 2, 1)), st4, {f"sa.b{i}": w4[f"b{i}"] for i in range(4)}, "sa", {})
    assert np.array_equal(nf4.transpose(1, 2).cpu().numpy(), onf4)


@pytest.mark.parametrize("overlap", [False, True])
def test_detector_tiny_end_to_end(orc, sad, dev, overlap):
    """3 SA stages -> size-adaptive cluster layer -> head, TINY topology, vs oracle and golden."""
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    cfg = config.TINY
    g = np.load(os.path.join(GOLDEN, "tiny_detector.npz"))
    w = synth.make_weights(cfg, 0)
    pts = synth.make_tiny_batch(0, 2, cfg.n_points)
    det = SADDetector(cfg, w, dev, overlap_fps=overlap)
    tr = {}
    boxes = det(_t(pts, dev), tr)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(tr["sa3"]["new_xyz"].cpu().numpy(),
                                  orc.gather_xyz(orc.gather_xyz(orc.gather_xyz(
                                      np.ascontiguousarray(pts[:, :, :3]), g["sa1_fps"]), g["sa2_fps"]), g["sa3_fps"]))
    _close(tr["sa3"]["out"].cpu().numpy(), g["sa3_out"], "sa3 features")
    rad = tr["cluster"]["radius"].cpu().numpy()
    _close(rad, g["radius"], "adaptive radius")
    if np.array_equal(rad, g["radius"]) and np.array_equal(tr["cluster"]["cand"].cpu().numpy(), g["cand"]):
        np.testing.assert_array_equal(tr["cluster"]["ball_idx"][0].cpu().numpy(), g["cl_idx0"])
        np.testing.assert_array_equal(tr["cluster"]["ball_idx"][1].cpu().numpy(), g["cl_idx1"])
    _close(tr["cluster"]["head"].cpu().numpy(), g["head"], "head output")
    b = boxes.cpu().numpy()
    assert b.shape == (2, cfg.n_cand, 9)
    np.testing.assert_array_equal(b[..., 8], g["boxes"][..., 8])
    _close(b, g["boxes"], "boxes")


def test_detector_kitti_stagewise(orc, sad, dev):
    """BASELINE configs[1] topology at full size (16384 pts): one scene, each GPU stage checked
    against the oracle fed with the GPU's own upstream tensors (so nothing compounds)."""
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    cfg = config.KITTI
    w = synth.make_weights(cfg, 0)
    pts = synth.make_batch(0, 1)
    det = SADDetector(cfg, w, dev)
    tr = {}
    boxes = det(_t(pts, dev), tr)
    torch.cuda.synchronize()
    xyz = np.ascontiguousarray(pts[:, :, :3])
    feat = np.ascontiguousarray(pts[:, :, 3:])
    for si, st in enumerate(cfg.stages):
        name = f"sa{si + 1}"
        otr = {}
        onx, onf = orc.sa_module(xyz, feat, st, w, name, otr)
        np.testing.assert_array_equal(tr[name]["new_xyz"].cpu().numpy(), onx)
        got = tr[name]["out"].cpu().numpy()
        _close(got, onf, f"{name} features (16384-pt scene)")
        xyz, feat = onx, got      # continue from the GPU's features
    assert boxes.shape == (1, cfg.n_cand, 9) and bool(torch.isfinite(boxes).all())


def test_random_shapes_grouped_mlp(orc, sad, dev):
    """25 random chains (1-4 layers, odd widths, any nsample 1..64, with/without features, counts
    from the ball query or derived from idx) vs the oracle: <= 1e-4 required, bit-exact observed."""
    from sad_amd import ops, synth
    rng = np.random.default_rng(77)
    exact = 0
    for trial in range(25):
        B = int(rng.integers(1, 3))
        N = int(rng.choice([300, 1000, 2500]))
        M = int(rng.integers(5, 200))
        S = int(rng.integers(1, 65))
        C = int(rng.choice([0, 1, 3, 4, 16, 37, 64]))
        L = int(rng.integers(1, 5))
        dims = [C + 3] + [int(rng.choice([8, 16, 24, 40, 64, 100, 128])) for _ in range(L)]
        xyz = rng.uniform(0, 2, (B, N, 3)).astype(np.float32)
        feat = rng.normal(size=(B, N, C)).astype(np.float32) if C else None
        new_xyz = orc.gather_xyz(xyz, orc.fps(xyz, M))
        r = float(rng.uniform(0.1, 0.8))
        layers = synth.make_mlp_weights(dims, rng)
        X, Cn = _t(xyz, dev), _t(new_xyz, dev)
        F = _t(feat, dev) if C else None
        idxs, cnts = ops.ball_query_multi((r,), (S,), X, Cn, return_counts=True)
        want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
        mlp = ops.PackedMLP(layers, True, dev)
        got1 = mlp.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
        got2 = mlp.grouped(X, F, Cn, idxs[0]).cpu().numpy()
        _close(got1, want, f"trial {trial} dims={dims} S={S} (counts from ball query)")
        assert np.array_equal(got1, got2), f"trial {trial}: counts from idx give a different result"
        exact += int(np.array_equal(got1, want))
    assert exact == 25, f"only {exact}/25 random chains were bit-exact"


def test_nested_fps_is_identity_prefix(orc, sad, dev):
    """The detector skips the FPS kernel for stages 2 and 3: fps(FPS-ordered points, M) must be
    0..M-1 (SPEC.md §2: start at index 0, ties -> lowest index), on KITTI-shaped scenes and on a
    lattice full of exact ties and duplicates; and the detector gives identical boxes either way."""
    import torch
    from sad_amd import config, ops, synth
    from sad_amd.detector import SADDetector
    xyz = np.ascontiguousarray(synth.make_batch(3, 2)[:, :, :3])
    ax = np.arange(16, dtype=np.float32)
    lat = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(1, 4096, 3).copy()
    lat[0, 500:600] = lat[0, 9]
    for pts, m1, m2 in ((xyz, 4096, 1024), (lat, 2048, 700)):
        X = _t(pts, dev)
        first = ops.gather_xyz(X, ops.fps(X, m1))
        second = ops.fps(first, m2).cpu().numpy()
        np.testing.assert_array_equal(second, np.tile(np.arange(m2, dtype=np.int32), (pts.shape[0], 1)))
        np.testing.assert_array_equal(orc.fps(first.cpu().numpy(), m2), second)
    cfg = config.TINY
    w = synth.make_weights(cfg, 0)
    p = _t(synth.make_tiny_batch(5, 2, cfg.n_points), dev)
    a = SADDetector(cfg, w, dev, nested_fps_shortcut=True)(p)
    b = SADDetector(cfg, w, dev, nested_fps_shortcut=False)(p)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_nuscenes_scale_sa_module(orc, sad, dev):
    """BASELINE configs[4] point count (65 536 points, fp32 here): one MSG stage 65 536 -> 2 048 through
    the large-N FPS kernel (global workspace), the grid ball query at N = 65 536 and the fused MLP,
    against the oracle."""
    from sad_amd import config, ops, synth
    from sad_amd.sa_module import SAModuleMSG
    pts = synth.make_batch(11, 1, 65536, extent=(-51.2, 51.2, -51.2, 51.2), n_boxes=120)
    xyz = np.ascontiguousarray(pts[:, :, :3])
    feat = np.ascontiguousarray(pts[:, :, 3:])
    st = config.SAStage(2048, (0.4, 1.0), (16, 32), ((16, 32), (16, 16, 32)), 32)
    rng = np.random.default_rng(4)
    w = {"b0": synth.make_mlp_weights([4, 16, 32], rng), "b1": synth.make_mlp_weights([4, 16, 16, 32], rng),
         "agg": synth.make_mlp_weights([64, 32], rng)}
    mod = SAModuleMSG(1, st, dev, w)
    nx, nf = mod.forward_pm(_t(xyz, dev), _t(feat, dev))
    ow = {"s.b0": w["b0"], "s.b1": w["b1"], "s.agg": w["agg"]}
    onx, onf = orc.sa_module(xyz, feat, st, ow, "s")
    np.testing.assert_array_equal(nx.cpu().numpy(), onx)
    _close(nf.cpu().numpy(), onf, "65536-point SA stage")


def test_multi_chain_dispatch_matches_single_launches(orc, sad, dev):
    """sad_mlp_chain_multi_f32 (the branches of a stage in one dispatch) writes the same bits as one
    launch per branch, including branches with different row blocking and a mixed-width stage."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(99)
    B, N, M, C = 2, 3000, 300, 16
    xyz = rng.random((B, N, 3), dtype=np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    radii, ns = [0.05, 0.1, 0.2], [16, 32, 64]
    mlps = [[32, 32, 64], [64, 64, 128], [64, 96, 128]]
    idxs, cnts = ops.ball_query_multi(radii, ns, _t(xyz, dev), _t(new_xyz, dev), return_counts=True)
    nets = [ops.PackedMLP(synth.make_mlp_weights([C + 3] + m, rng), True, dev, name=f"t.b{i}") for i, m in enumerate(mlps)]
    width = sum(m[-1] for m in mlps)
    single = torch.zeros(B, M, width, device=dev)
    merged = torch.zeros(B, M, width, device=dev)
    calls, off = [], 0
    for net, idx, cnt in zip(nets, idxs, cnts):
        net.grouped(_t(xyz, dev), _t(feat, dev), _t(new_xyz, dev), idx, out=single, col_off=off, cnt=cnt)
        calls.append((net, _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev), idx, merged, off, cnt))
        off += net.out_channels
    ops.grouped_multi(calls)
    assert torch.equal(single, merged)


def test_every_geometry_code_is_bit_identical(orc, sad, dev):
    """Every workgroup geometry the autotuner may pick (wave grid, row blocking, flexible item
    distribution, groups per workgroup, global / per-workgroup packing, VALU kernel) gives the
    oracle's bits; geometries that do not fit LDS are refused with SAD_EUNSUPPORTED, never wrong."""
    import torch
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(4711)
    B, N, M, S, C = 2, 2500, 200, 32, 64
    xyz = rng.random((B, N, 3), dtype=np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    idxs, cnts = ops.ball_query_multi([0.12], [S], _t(xyz, dev), _t(new_xyz, dev), return_counts=True)
    layers = synth.make_mlp_weights([C + 3, 64, 96, 128], rng)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
    net = ops.PackedMLP(layers, True, dev)
    rows = rng.normal(size=(700, 384)).astype(np.float32)
    play = synth.make_mlp_weights([384, 128, 40], rng)
    pwant = orc.mlp_rows(rows, play)
    pnet = ops.PackedMLP(play, False, dev)
    codes = list(ops.PackedMLP._CANDIDATES)
    codes += [c + 1000 * f for c in (801, 811, 100811) for f in ops.PackedMLP._F_CODES]
    codes += [c + 10000 * d for c in (801, 821, 100821, 5811) for d in (1, 2)]
    codes += [2, 4]        # register-resident chain and its cooperative variant (this chain is the SA2 shape 67 -> 64 -> 96 -> 128)
    ran, special = 0, []
    try:
        for code in codes:
            _lib.set_option("mlp_force", code)
            try:
                got = net.grouped(_t(xyz, dev), _t(feat, dev), _t(new_xyz, dev), idxs[0], cnt=cnts[0]).cpu().numpy()
            except RuntimeError as e:
                assert "(-2)" in str(e), f"geometry {code}: {e}"      # SAD_EUNSUPPORTED only
                continue
            assert np.array_equal(got, want), f"grouped geometry {code}: max diff {np.abs(got - want).max()}"
            if code in (2, 4):
                special.append(code)
            if code % 100000 < 1000:      # every wave-grid / flag code also on plain rows (f / packing codes are grouped-only)
                try:
                    pg = pnet.rows(_t(rows, dev)).cpu().numpy()
                    assert np.array_equal(pg, pwant), f"plain geometry {code}"
                except RuntimeError as e:
                    assert "(-2)" in str(e), f"plain geometry {code}: {e}"
            ran += 1
    finally:
        _lib.set_option("mlp_force", 0)
    assert ran >= 20, f"only {ran} geometries ran"
    assert special == [2, 4], f"register-resident geometries that ran: {special}"


@pytest.mark.parametrize("mlp", [[16, 16, 32], [32, 32, 64]])
def test_valu_kernel_parity(orc, sad, dev, mlp):
    """geometry 1 = the row-per-lane vector-ALU kernel for the two narrow SA1 chain shapes (an
    autotune candidate): same fmaf chains, bit-identical to the oracle, with and without counts."""
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(sum(mlp))
    B, N, M, S = 2, 3000, 400, 32 if mlp[0] == 16 else 64
    xyz = rng.random((B, N, 3), dtype=np.float32)
    feat = rng.random((B, N, 1), dtype=np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    idxs, cnts = ops.ball_query_multi([0.08], [S], _t(xyz, dev), _t(new_xyz, dev), return_counts=True)
    layers = synth.make_mlp_weights([4] + mlp, rng)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
    net = ops.PackedMLP(layers, True, dev)
    _lib.set_option("mlp_force", 1)
    try:
        got = net.grouped(_t(xyz, dev), _t(feat, dev), _t(new_xyz, dev), idxs[0], cnt=cnts[0]).cpu().numpy()
        got2 = net.grouped(_t(xyz, dev), _t(feat, dev), _t(new_xyz, dev), idxs[0]).cpu().numpy()
    finally:
        _lib.set_option("mlp_force", 0)
    assert np.array_equal(got, want) and np.array_equal(got2, want)


REG_CASES = [
    # (B, N, M, S, C, mlp, radius) — every compiled shape of the register-resident chain kernel (geometry 2)
    (2, 3000, 700, 32, 1, [16, 16, 32], 0.08),        # SA1 narrow branch (single feature channel, strided view)
    (2, 3000, 700, 64, 1, [32, 32, 64], 0.15),        # SA1 wide branch, nsample 64
    (2, 3000, 500, 32, 4, [16, 16, 32], 0.1),         # nuScenes SA1 (4 feature channels: one 16-byte chunk)
    (1, 1024, 256, 32, 0, [64, 64, 128], 0.2),        # BASELINE configs[0] (no features)
    (2, 2000, 400, 32, 64, [64, 64, 128], 0.2),       # SA2
    (2, 2000, 400, 64, 64, [64, 96, 128], 0.35),
    (2, 1024, 300, 32, 128, [128, 128, 256], 0.3),    # SA3
    (2, 1024, 300, 32, 128, [128, 192, 256], 0.5),
    (2, 1024, 300, 32, 128, [128, 256, 256], 0.7),
    (1, 200, 19, 8, 1, [16, 16, 32], 0.5),            # fewer rows than one tile
]


@pytest.mark.parametrize("B,N,M,S,C,mlp,r", REG_CASES)
def test_register_chain_kernel_parity(orc, sad, dev, B, N, M, S, C, mlp, r):
    """geometry 2 = one wave carries a 32-row tile through the whole chain in registers (csrc/mlp_reg.hip):
    bit-identical to the oracle's fmaf chains and to the tiled kernel, with ragged groups, groups that
    straddle tiles (atomic max merge) and the padding skip both on and off."""
    import torch
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(N + M + S + C + sum(mlp))
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    pts4 = rng.uniform(0, 1, (B, N, 4)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, Cn = _t(xyz, dev), _t(new_xyz, dev)
    if C == 1:        # the detector's SA1 input: intensity as a strided view of [B,N,4]
        P = _t(pts4, dev)
        F = P[:, :, 3:]
        feat = np.ascontiguousarray(pts4[:, :, 3:])
    elif C:
        feat = rng.normal(size=(B, N, C)).astype(np.float32)
        F = _t(feat, dev)
    else:
        feat, F = None, None
    idxs, cnts = ops.ball_query_multi((r,), (S,), X, Cn, return_counts=True)
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
    net = ops.PackedMLP(layers, True, dev)
    tiled = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
    assert np.array_equal(tiled, want)
    _lib.set_option("mlp_force", 2)
    try:
        got = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
        _lib.set_option("mlp_nodedup", 1)
        dense = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
    finally:
        _lib.set_option("mlp_force", 0)
        _lib.set_option("mlp_nodedup", 0)
    rows = int(cnts[0].clamp(min=1).sum().item())
    print(f"[parity] register chain {[C + 3] + mlp} S={S}: {rows} packed rows, bit-exact={np.array_equal(got, want)}")
    assert np.array_equal(got, want), f"max diff {np.abs(got - want).max():.3e}"
    assert np.array_equal(dense, want), "padding skip off: different result"


COOP_CASES = [
    # (B, N, M, S, C, mlp, radius) — geometry 4: the cooperative register-resident chain (SA3 and SA2 shapes)
    (2, 1024, 300, 32, 128, [128, 128, 256], 0.3),
    (2, 1024, 300, 32, 128, [128, 192, 256], 0.5),
    (2, 1024, 300, 32, 128, [128, 256, 256], 0.7),
    (1, 300, 7, 16, 128, [128, 256, 256], 0.2),       # fewer rows than one work item (dummy waves)
    (3, 2048, 512, 32, 128, [128, 128, 256], 0.9),    # full groups: many tiles per workgroup
    (2, 2000, 400, 32, 64, [64, 64, 128], 0.2),       # SA2
    (2, 2000, 400, 64, 64, [64, 96, 128], 0.35),      # SA2, layer-2 tiles of 12 k-groups (not whole stages)
    (1, 500, 40, 32, 64, [64, 96, 128], 0.9),
]


@pytest.mark.parametrize("B,N,M,S,C,mlp,r", COOP_CASES)
def test_cooperative_chain_kernel_parity(orc, sad, dev, B, N, M, S, C, mlp, r):
    """geometry 4 = four waves carry four tiles through the chain in registers and share the weight stream through
    an LDS ring (csrc/mlp_coop.hip): bit-identical to the oracle and to the tiled kernel, alone, with the padding
    skip off, and as the merged three-chain dispatch of a stage."""
    import torch
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(N + M + S + C + sum(mlp))
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, F, Cn = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    idxs, cnts = ops.ball_query_multi((r,), (S,), X, Cn, return_counts=True)
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
    net = ops.PackedMLP(layers, True, dev)
    net.default_geometry = 4
    got = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
    rows = int(cnts[0].clamp(min=1).sum().item())
    print(f"[parity] cooperative chain {[C + 3] + mlp} S={S}: {rows} packed rows, bit-exact={np.array_equal(got, want)}")
    assert np.array_equal(got, want), f"max diff {np.abs(got - want).max():.3e}"
    _lib.set_option("mlp_nodedup", 1)
    try:
        dense = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
    finally:
        _lib.set_option("mlp_nodedup", 0)
    assert np.array_equal(dense, want), "padding skip off: different result"


def test_cooperative_three_chain_dispatch(orc, sad, dev):
    """The three SA3 branches as ONE cooperative dispatch (sad_mlp_chain_multi_f32): chains of different shapes
    follow each other in a workgroup's item list, the ring carries over from one chain's stream to the next."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(77)
    B, N, M, C = 2, 1024, 384, 128
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, F, Cn = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    radii, ns = (0.2, 0.35, 0.5), (32, 32, 32)
    mlps = ([128, 128, 256], [128, 192, 256], [128, 256, 256])
    idxs, cnts = ops.ball_query_multi(radii, ns, X, Cn, return_counts=True)
    out = torch.zeros((B, M, 768), device=dev)
    calls, wants = [], []
    for bi, mlp in enumerate(mlps):
        layers = synth.make_mlp_weights([C + 3] + mlp, rng)
        net = ops.PackedMLP(layers, True, dev)
        net.default_geometry = 4
        calls.append((net, X, F, Cn, idxs[bi], out, 256 * bi, cnts[bi]))
        wants.append(orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[bi].cpu().numpy(), layers))
    ops.grouped_multi(calls)
    got = out.cpu().numpy()
    for bi in range(3):
        assert np.array_equal(got[:, :, 256 * bi:256 * (bi + 1)], wants[bi]), f"branch {bi}"


def test_cooperative_steal_refill_is_exercised(orc, sad, dev):
    """the cross-queue steal + weight-ring refill branch of mlp_coop_kernel (csrc/mlp_coop.hip; the
    intermittent wrong-rows bug fixed in 4c2dbaa lived there) normally runs only at the tail of a dispatch, when a
    workgroup whose own queue is empty takes an item of another chain than the one whose first stages it prefetched.
    ``mlp_steal_after`` makes every workgroup take its items from the OTHER queues from the start (v = 1) or after
    two own items (v = 3), one at a time, so the branch runs at every chain change of every workgroup; the refill
    counter in the table header proves it ran.  Three chains of different shapes, bit-exact vs the oracle (SPEC.md §6)."""
    import torch
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(4242)
    B, N, M, C = 4, 1024, 512, 128
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, F, Cn = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    radii, ns = (0.45, 0.6, 0.8), (32, 32, 32)          # mostly full groups: ~500 work items per chain
    mlps = ([128, 128, 256], [128, 192, 256], [128, 256, 256])
    idxs, cnts = ops.ball_query_multi(radii, ns, X, Cn, return_counts=True)
    nets, wants = [], []
    for bi, mlp in enumerate(mlps):
        layers = synth.make_mlp_weights([C + 3] + mlp, rng)
        net = ops.PackedMLP(layers, True, dev)
        net.default_geometry = 4
        nets.append(net)
        wants.append(orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[bi].cpu().numpy(), layers))
    want = np.concatenate(wants, axis=2)
    items = [int(-(-int(c.clamp(min=1).sum().item()) // 128)) for c in cnts]

    def run(wss):
        out = torch.zeros((B, M, 768), device=dev)
        ops.grouped_multi([(nets[bi], X, F, Cn, idxs[bi], out, 256 * bi, cnts[bi], wss[bi]) for bi in range(3)])
        return out.cpu().numpy()

    def refills(wss):
        return sum(ops.workspace_status(w)["refills"] for w in wss)

    wss = ops.rowscan_multi(idxs, cnts, N)
    base = run(wss)
    assert np.array_equal(base, want), "plain dispatch"
    r_plain = refills(wss)
    _lib.set_option("mlp_dyn_slots", 1)                  # 256 workgroups: several items each
    try:
        for v in (1, 3):
            _lib.set_option("mlp_steal_after", v)
            wss = ops.rowscan_multi(idxs, cnts, N)
            for rep in range(3):                         # the same tables again: the queues are re-armed in this mode too
                got = run(wss)
                assert np.array_equal(got, want), f"mlp_steal_after={v}, launch {rep}: max diff {np.abs(got - want).max():.3e}"
            r = refills(wss)
            print(f"[steal/refill] items per chain {items}, mlp_steal_after={v}: {r} ring refills in 3 launches "
                  f"(plain dispatch: {r_plain}), bit-exact")
            assert r >= 3 * 64, f"mlp_steal_after={v}: the refill branch ran only {r} times"
    finally:
        _lib.set_option("mlp_steal_after", 0)
        _lib.set_option("mlp_dyn_slots", 0)


def test_workspace_in_use_marker(orc, sad, dev):
    """"one dispatch at a time per row-packing workspace" is checked, not only documented: with
    ``mlp_check_inuse`` a dispatch that finds another dispatch's id in the queue header raises the conflict flag
    (forged here, so the test does not depend on two launches really overlapping); an undisturbed dispatch leaves the
    header clean and releases it."""
    import torch
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(17)
    B, N, M, C = 2, 1024, 512, 128
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    X, F = _t(xyz, dev), _t(feat, dev)
    Cn = X[:, :M].contiguous()
    idxs, cnts = ops.ball_query_multi((0.3,), (32,), X, Cn, return_counts=True)
    net = ops.PackedMLP(synth.make_mlp_weights([C + 3, 128, 128, 256], rng), True, dev)
    net.default_geometry = 4
    _lib.set_option("mlp_check_inuse", 1)
    try:
        ws = ops.rowscan_multi(idxs, cnts, N)[0]
        ref = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0], ws=ws)
        st = ops.workspace_status(ws)
        assert st["conflict"] == 0 and st["in_use"] == 0, st
        ops.check_workspace(ws)
        ws.view(torch.int32)[_lib.WS_INUSE] = 0x12345          # "another dispatch owns these queues"
        net.grouped(X, F, Cn, idxs[0], cnt=cnts[0], ws=ws)
        st = ops.workspace_status(ws)
        assert st["conflict"] == 1, st
        with pytest.raises(RuntimeError, match="two dispatches"):
            ops.check_workspace(ws)
        assert st["in_use"] == 0, "the last workgroup out releases the marker"
        ws2 = ops.rowscan_multi(idxs, cnts, N)[0]              # a fresh table: clean again, same result
        again = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0], ws=ws2)
        assert ops.workspace_status(ws2)["conflict"] == 0 and torch.equal(again, ref)
    finally:
        _lib.set_option("mlp_check_inuse", 0)


@pytest.mark.parametrize("steal_after", [0, 1])
def test_cooperative_dispatch_rearms_its_item_queues(orc, sad, dev, steal_after):
    """The cooperative kernel pulls its work items from per-XCD queues in the header of the row-packing table and
    the last workgroup out re-arms them: the same prescanned tables serve any number of launches (a queue left
    exhausted would make the next launch compute nothing), also when two streams run dispatches side by side on
    tables of their own.  ``steal_after=1``: the same with every item taken from another XCD's queue (the steal /
    refill path, see test_cooperative_steal_refill_is_exercised)."""
    import torch
    from sad_amd import _lib, ops, synth
    _lib.set_option("mlp_steal_after", steal_after)
    try:
        _rearm_body(orc, dev, torch, ops, synth)
    finally:
        _lib.set_option("mlp_steal_after", 0)


def _rearm_body(orc, dev, torch, ops, synth):
    rng = np.random.default_rng(99)
    B, N, M, C = 4, 1024, 512, 128
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, F, Cn = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    radii, ns = (0.15, 0.3), (32, 32)
    mlps = ([128, 128, 256], [128, 256, 256])
    idxs, cnts = ops.ball_query_multi(radii, ns, X, Cn, return_counts=True)
    nets, wants = [], []
    for bi, mlp in enumerate(mlps):
        layers = synth.make_mlp_weights([C + 3] + mlp, rng)
        net = ops.PackedMLP(layers, True, dev)
        net.default_geometry = 4
        nets.append(net)
        wants.append(orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[bi].cpu().numpy(), layers))
    want = np.concatenate(wants, axis=2)

    def dispatch(wss, out):
        ops.grouped_multi([(nets[bi], X, F, Cn, idxs[bi], out, 256 * bi, cnts[bi], wss[bi]) for bi in range(2)])

    import os
    stress = int(os.environ.get("SAD_STRESS", "1"))          # SAD_STRESS=20: a longer soak of the same checks
    wss = ops.rowscan_multi(idxs, cnts, N)
    for rep in range(6 * stress):                            # one table, many launches
        out = torch.zeros((B, M, 512), device=dev)
        dispatch(wss, out)
        assert np.array_equal(out.cpu().numpy(), want), f"launch {rep} on the same tables"
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    wa, wb = ops.rowscan_multi(idxs, cnts, N), ops.rowscan_multi(idxs, cnts, N)
    torch.cuda.synchronize()
    outs = []
    for rep in range(4 * stress):                            # two streams, a table set each
        oa, ob = torch.zeros((B, M, 512), device=dev), torch.zeros((B, M, 512), device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            dispatch(wa, oa)
        with torch.cuda.stream(s2):
            dispatch(wb, ob)
        torch.cuda.synchronize()
        outs += [oa, ob]
    for i, o in enumerate(outs):
        assert np.array_equal(o.cpu().numpy(), want), f"concurrent dispatch {i}"


def test_register_chain_refuses_other_shapes(orc, sad, dev):
    """A chain without a compiled shape is refused with SAD_EUNSUPPORTED (autotuners skip it), never wrong."""
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(5)
    xyz = rng.uniform(0, 1, (1, 500, 3)).astype(np.float32)
    X = _t(xyz, dev)
    Cn = X[:, :50].contiguous()
    idxs, cnts = ops.ball_query_multi((0.3,), (16,), X, Cn, return_counts=True)
    net = ops.PackedMLP(synth.make_mlp_weights([3, 24, 40], rng), True, dev)
    narrow = ops.PackedMLP(synth.make_mlp_weights([3, 16, 16, 32], rng), True, dev)   # an SA1 shape: geometry 2 yes, 4 no
    for code, chain in ((2, net), (4, net), (4, narrow)):
        _lib.set_option("mlp_force", code)
        try:
            with pytest.raises(RuntimeError, match=r"\(-2\)"):
                chain.grouped(X, None, Cn, idxs[0], cnt=cnts[0])
        finally:
            _lib.set_option("mlp_force", 0)


LAYERED_CASES = [
    # (B, N, M, S, C, mlp, radius) — geometry 3: layer-streamed chain (every padded width a multiple of 128)
    (2, 512, 256, 16, 256, [256, 256, 512], 0.25),    # cluster branch 0
    (2, 512, 256, 32, 256, [256, 512, 1024], 0.35),   # cluster branch 1
    (1, 300, 70, 32, 128, [128, 256], 0.4),           # two layers
    (1, 300, 70, 24, 8, [256], 0.4),                  # one layer, nsample 24, narrow input
    (1, 400, 33, 32, 64, [128, 128, 128, 256], 0.5),  # four layers
]


@pytest.mark.parametrize("B,N,M,S,C,mlp,r", LAYERED_CASES)
def test_layer_streamed_chain_parity(orc, sad, dev, B, N, M, S, C, mlp, r):
    """geometry 3 = one launch per layer, (32-row tile x 128 channels) work items, activations between layers in
    scratch (csrc/mlp_layer.hip): bit-identical to the oracle and to the tiled kernel, alone and as the merged
    two-chain dispatch the cluster layer uses."""
    import torch
    from sad_amd import _lib, ops, synth
    rng = np.random.default_rng(N + M + S + C + sum(mlp))
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, F, Cn = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    idxs, cnts = ops.ball_query_multi((r,), (S,), X, Cn, return_counts=True)
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
    net = ops.PackedMLP(layers, True, dev)
    assert net._layered_ok
    net.default_geometry = 3
    got = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
    rows = int(cnts[0].clamp(min=1).sum().item())
    print(f"[parity] layer-streamed chain {[C + 3] + mlp} S={S}: {rows} packed rows, bit-exact={np.array_equal(got, want)}")
    assert np.array_equal(got, want), f"max diff {np.abs(got - want).max():.3e}"
    _lib.set_option("mlp_nodedup", 1)
    try:
        dense = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
    finally:
        _lib.set_option("mlp_nodedup", 0)
    assert np.array_equal(dense, want), "padding skip off: different result"
    # knob mlp_layer_queue: the same items pulled from the launching stream's per-XCD queues (twice: the last workgroup of a
    # launch re-arms the queues for the next one)
    _lib.set_option("mlp_layer_queue", 1)
    try:
        for _ in range(2):
            queued = net.grouped(X, F, Cn, idxs[0], cnt=cnts[0]).cpu().numpy()
            assert np.array_equal(queued, want), "queue-fed items: different result"
    finally:
        _lib.set_option("mlp_layer_queue", 0)


PLAIN_LAYERED_CASES = [
    # (rows, dims, relu_mask, ld_out, col_off) — geometry 3 on plain rows (the stage aggregations of the detector)
    (777, [320, 128], None, 128, 0),          # sa2.agg, ragged last block
    (1000, [768, 256], None, 256, 0),         # sa3.agg
    (4096, [768, 256], None, 384, 128),       # into a slice of a wider buffer
    (333, [128, 128, 256], 0b01, 256, 0),     # two layers through scratch, last one linear
    (130, [8, 256, 128, 128], 0b101, 128, 0), # three layers
    (1, [64, 128], None, 128, 0),             # a single row
]


@pytest.mark.parametrize("rows,dims,mask,ld_out,col_off", PLAIN_LAYERED_CASES)
def test_layer_streamed_plain_rows_parity(orc, sad, dev, rows, dims, mask, ld_out, col_off):
    """geometry 3 on plain rows: every layer one GEMM launch of (128 rows x 128 channels) items, bit-identical to
    the oracle's fmaf chains; columns outside the output slice are left alone."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(rows + sum(dims))
    layers = synth.make_mlp_weights(dims, rng)
    x = rng.normal(size=(rows, dims[0])).astype(np.float32)
    want = orc.mlp_rows(x, layers, relu_mask=mask)
    net = ops.PackedMLP(layers, False, dev, relu_mask=mask)
    assert net._layered_ok
    net.default_geometry = 3
    out = torch.full((rows, ld_out), -7.0, dtype=torch.float32, device=dev)
    net.rows(_t(x, dev), out=out, col_off=col_off)
    got = out.cpu().numpy()
    assert np.array_equal(got[:, col_off:col_off + dims[-1]], want), f"max diff {np.abs(got[:, col_off:col_off + dims[-1]] - want).max():.3e}"
    rest = np.delete(got, np.s_[col_off:col_off + dims[-1]], axis=1)
    assert (rest == -7.0).all(), "columns outside the slice were written"
    from sad_amd import _lib
    _lib.set_option("mlp_layer_queue", 1)               # (plain rows have no row-packing table: the queues are the stream's)
    try:
        out2 = torch.full((rows, ld_out), -7.0, dtype=torch.float32, device=dev)
        net.rows(_t(x, dev), out=out2, col_off=col_off)
        assert np.array_equal(out2.cpu().numpy(), got), "queue-fed items: different result"
    finally:
        _lib.set_option("mlp_layer_queue", 0)


ROWS_LAYER_CASES = [
    # (rows, C, C_out, relu, ld_out, col_off) — geometry 5: the row-streaming plain layer (csrc/mlp_rows.hip)
    (4096, 128, 64, True, 64, 0),            # sa1.agg
    (777, 384, 128, True, 128, 0),           # sa2.agg, ragged last block
    (1000, 768, 256, True, 384, 128),        # sa3.agg into a slice of a wider buffer
    (130, 1536, 512, True, 512, 0),          # cluster.agg: four channel blocks per row block
    (333, 256, 10, False, 10, 0),            # a linear head layer: C_out not a multiple of 4, unaligned rows of the output
    (1, 8, 33, True, 40, 3),                 # a single row, one k-group, odd column offset
    (257, 40, 96, True, 96, 0),              # five k-groups (a partial last chunk), three channel tiles
]


@pytest.mark.parametrize("rows,C,cout,relu,ld_out,col_off", ROWS_LAYER_CASES)
def test_row_streaming_layer_parity(orc, sad, dev, rows, C, cout, relu, ld_out, col_off):
    """geometry 5: one plain layer, a wave owns 32 rows and every output channel of its item; bit-identical to the oracle's
    fmaf chains (SPEC.md §6) and to the tiled kernel; columns outside the output slice are left alone."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(rows + C + cout)
    layers = synth.make_mlp_weights([C, cout], rng)
    x = rng.normal(size=(rows, C)).astype(np.float32)
    mask = 1 if relu else 0
    want = orc.mlp_rows(x, layers, relu_mask=mask)
    net = ops.PackedMLP(layers, False, dev, relu_mask=mask)
    tiled = net.rows(_t(x, dev)).cpu().numpy()
    assert np.array_equal(tiled, want)
    net.default_geometry = 5
    out = torch.full((rows, ld_out), -7.0, dtype=torch.float32, device=dev)
    net.rows(_t(x, dev), out=out, col_off=col_off)
    got = out.cpu().numpy()
    assert np.array_equal(got[:, col_off:col_off + cout], want), f"max diff {np.abs(got[:, col_off:col_off + cout] - want).max():.3e}"
    rest = np.delete(got, np.s_[col_off:col_off + cout], axis=1)
    assert (rest == -7.0).all(), "columns outside the slice were written"


def test_row_streaming_layer_refusal(sad, dev):
    """Chains geometry 5 cannot take (two layers, C not a multiple of 8) are refused with SAD_EUNSUPPORTED."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(6)
    for dims in ([128, 64, 32], [132, 64]):
        net = ops.PackedMLP(synth.make_mlp_weights(dims, rng), False, dev)
        net.default_geometry = 5
        with pytest.raises(RuntimeError, match=r"\(-2\)"):
            net.rows(torch.zeros((64, dims[0]), device=dev))


def test_layer_streamed_plain_rows_refusal(sad, dev):
    """Shapes the plain layer-streamed path cannot take are refused with SAD_EUNSUPPORTED, never computed wrong."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(5)
    for dims in ([128, 64], [132, 128], [256, 128, 6]):
        net = ops.PackedMLP(synth.make_mlp_weights(dims, rng), False, dev, relu_mask=(1 << (len(dims) - 2)) - 1 if len(dims) > 2 else None)
        net.default_geometry = 3
        with pytest.raises(RuntimeError, match=r"\(-2\)"):
            net.rows(torch.zeros((64, dims[0]), device=dev))


def test_layer_streamed_two_chain_dispatch(orc, sad, dev):
    """The cluster layer's two branches as ONE sequence of per-layer launches (sad_mlp_chain_multi_f32)."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(321)
    B, N, M, C = 2, 512, 256, 256
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = np.maximum(rng.normal(size=(B, N, C)).astype(np.float32), 0)
    new_xyz = np.ascontiguousarray(xyz[:, :M] + 0.01)
    rad = rng.uniform(0.15, 0.3, (B, M)).astype(np.float32)
    X, F, Cn = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    idxs, cnts = ops.ball_query_multi((1.0, 2.0), (16, 32), X, Cn, _t(rad, dev), return_counts=True)
    mlps = [[256, 256, 512], [256, 512, 1024]]
    nets = [ops.PackedMLP(synth.make_mlp_weights([C + 3] + m, rng), True, dev, name=f"c.b{i}") for i, m in enumerate(mlps)]
    width = sum(m[-1] for m in mlps)
    merged = torch.zeros(B, M, width, device=dev)
    single = torch.zeros(B, M, width, device=dev)
    calls, off = [], 0
    for net, idx, cnt in zip(nets, idxs, cnts):
        net.default_geometry = 0
        net.grouped(X, F, Cn, idx, out=single, col_off=off, cnt=cnt)       # tiled kernel
        net.default_geometry = 3
        calls.append((net, X, F, Cn, idx, merged, off, cnt))
        off += net.out_channels
    ops.grouped_multi(calls)
    assert torch.equal(single, merged)


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [2, 3, 4])
def test_scan_prepares_uninitialised_pooling_buffers(orc, sad, dev, geom):
    """sad_mlp_rowscan_init: the scan zero-fills exactly the output rows the chain kernels combine with an atomic max, so
    the pooling buffer may hold anything on entry (here NaN, and a neighbouring column slice that must stay untouched) —
    prescanned tables and the dispatch's own scan alike, every table-driven geometry; bit-exact vs the oracle."""
    import torch
    from sad_amd import ops, synth
    B, N, M, S, C = 2, 1024, 512, 32, 128
    mlp = [128, 128, 256] if geom != 3 else [128, 256, 256]
    rng = np.random.default_rng(geom)
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, F, Cn = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    idxs, cnts = ops.ball_query_multi((0.12,), (S,), X, Cn, return_counts=True)
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
    net = ops.PackedMLP(layers, True, dev)
    net.default_geometry = geom
    co, off, ld = mlp[-1], 64, mlp[-1] + 96
    straddling = int(((cnts[0].clamp(min=1).flatten().cumsum(0) - 1) // 32 != (cnts[0].clamp(min=1).flatten().cumsum(0) - cnts[0].clamp(min=1).flatten()) // 32).sum())
    assert straddling > 50, "the case must have groups that straddle tiles"
    for prescan in (True, False):
        out = torch.full((B, M, ld), float("nan"), dtype=torch.float32, device=dev)
        out[:, :, :off] = -5.0
        out[:, :, off + co:] = -6.0
        ws = ops.rowscan_multi(idxs, cnts, N, outs=[(out, off, co)])[0] if prescan else None
        net.grouped(X, F, Cn, idxs[0], out=out, col_off=off, cnt=cnts[0], ws=ws)
        got = out.cpu().numpy()
        assert np.array_equal(got[:, :, off:off + co], want), f"prescan={prescan}: max diff {np.nanmax(np.abs(got[:, :, off:off + co] - want))}"
        assert (got[:, :, :off] == -5.0).all() and (got[:, :, off + co:] == -6.0).all(), "columns outside the slice were written"


def test_scan_prepares_unaligned_output_slice(orc, sad, dev):
    """sad_mlp_rowscan_init on an output slice that is not 16-byte aligned (odd column offset, odd row stride): the scalar
    zero-fill path; a chain whose last width is not a multiple of 4."""
    import torch
    from sad_amd import ops, synth
    B, N, M, S, C = 1, 512, 256, 16, 1
    mlp = [16, 16, 32]
    rng = np.random.default_rng(7)
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, F, Cn = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    idxs, cnts = ops.ball_query_multi((0.15,), (S,), X, Cn, return_counts=True)
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
    net = ops.PackedMLP(layers, True, dev)
    net.default_geometry = 2
    co, off, ld = mlp[-1], 3, mlp[-1] + 9
    out = torch.full((B, M, ld), float("nan"), dtype=torch.float32, device=dev)
    out[:, :, :off] = -5.0
    out[:, :, off + co:] = -6.0
    ws = ops.rowscan_multi(idxs, cnts, N, outs=[(out, off, co)])[0]
    net.grouped(X, F, Cn, idxs[0], out=out, col_off=off, cnt=cnts[0], ws=ws)
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :, off:off + co], want)
    assert (got[:, :, :off] == -5.0).all() and (got[:, :, off + co:] == -6.0).all()


def test_layer_queue_on_two_streams_at_once(orc, sad, dev):
    """mlp_layer_queue=1: the item queues belong to the launching stream, so layer-streamed chains launched on two
    streams at the same time do not share counters; both give the oracle's bits, several rounds."""
    import torch
    from sad_amd import _lib, ops, synth
    B, N, M, S, C = 2, 1024, 512, 32, 128
    mlp = [128, 256, 256]
    rng = np.random.default_rng(11)
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, :M])
    X, F, Cn = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    idxs, cnts = ops.ball_query_multi((0.12,), (S,), X, Cn, return_counts=True)
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idxs[0].cpu().numpy(), layers)
    nets = [ops.PackedMLP(layers, True, dev) for _ in range(2)]
    for n in nets:
        n.default_geometry = 3
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    _lib.set_option("mlp_layer_queue", 1)
    try:
        for _ in range(4):
            outs = []
            for n, st in zip(nets, streams):
                with torch.cuda.stream(st):
                    outs.append(n.grouped(X, F, Cn, idxs[0], cnt=cnts[0]))
            torch.cuda.synchronize()
            for o in outs:
                assert np.array_equal(o.cpu().numpy(), want)
    finally:
        _lib.set_option("mlp_layer_queue", 0)



@pytest.mark.parametrize("S,C,mlp", [(32, 6, [32, 32, 64]), (32, 32, [64, 96, 128]), (64, 64, [64, 128]), (32, 96, [128, 196, 256]),
                                     (32, 128, [128, 128, 256, 256]), (32, 256, [256, 384, 512]), (64, 13, [40, 72])])
def test_uncompiled_chains_untuned_path(orc, sad, dev, S, C, mlp):
    """Drop-in generality: chains that are not compiled shapes, called with counts and NOT tuned, run on
    the kernel the library prefers for them — never 0 for a valid chain (round 5: a code of the tiled kernel, or the
    layer-streamed chain when every width is a multiple of 128) — and give the oracle's pooled features bit for bit."""
    from sad_amd import ops, synth
    rng = np.random.default_rng(S + 3 * C + sum(mlp))
    B, N, M = 2, 1500, 300
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = orc.gather_xyz(xyz, orc.fps(xyz, M))
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    net = ops.PackedMLP(layers, True, dev)
    assert net.preferred_geometry != 0
    X, F, NX = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    (idx,), (cnt,) = ops.ball_query_multi([0.17], [S], X, NX, return_counts=True)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idx.cpu().numpy(), layers, skip_padding=True)
    import torch
    out = torch.zeros((B, M, mlp[-1]), device=dev)
    got = net.grouped(X, F, NX, idx, out=out, cnt=cnt).cpu().numpy()
    assert np.array_equal(got, want), f"C={C} {mlp}: un-tuned geometry {net.preferred_geometry} differs from the oracle"


@pytest.mark.parametrize("S,C,mlp", [(32, 32, [64, 96, 128]), (32, 128, [100, 100, 200]), (32, 96, [128, 196, 256]), (64, 60, [64, 64, 120])])
def test_zero_padded_chain_runs_on_the_dominating_compiled_shape(orc, sad, dev, S, C, mlp):
    """A chain that is not a compiled shape but is dominated by one is packed zero-padded onto it (sad_mlp_padded_dims,
    sad_mlp_args.c_out: ABI 3) and runs on the register-resident / cooperative kernels: the oracle's pooled features bit for
    bit, with and without counts, into a slice of a wider buffer whose other columns stay untouched."""
    import torch
    from sad_amd import ops, synth
    rng = np.random.default_rng(S + 5 * C + sum(mlp))
    B, N, M = 2, 1500, 300
    xyz = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
    feat = rng.normal(size=(B, N, C)).astype(np.float32)
    new_xyz = orc.gather_xyz(xyz, orc.fps(xyz, M))
    layers = synth.make_mlp_weights([C + 3] + mlp, rng)
    net = ops.PackedMLP(layers, True, dev)
    assert net.padded and net.pack_dims != net.dims and net.preferred_geometry in (2, 4), (net.pack_dims, net.preferred_geometry)
    X, F, NX = _t(xyz, dev), _t(feat, dev), _t(new_xyz, dev)
    (idx,), (cnt,) = ops.ball_query_multi([0.2], [S], X, NX, return_counts=True)
    want = orc.sa_group_mlp_max(xyz, feat, new_xyz, idx.cpu().numpy(), layers, skip_padding=True)
    co = mlp[-1]
    buf = torch.full((B, M, co + 24), -3.0, device=dev)
    buf[:, :, 8:8 + co] = 0.0
    got = net.grouped(X, F, NX, idx, out=buf, col_off=8, cnt=cnt).cpu().numpy()
    assert np.array_equal(got[:, :, 8:8 + co], want), f"{mlp}: padded chain differs from the oracle"
    assert (got[:, :, :8] == -3).all() and (got[:, :, 8 + co:] == -3).all(), "columns outside the slice were written"
    got2 = net.grouped(X, F, NX, idx).cpu().numpy()               # no counts: derived from idx on the host side
    assert np.array_equal(got2, want)
    for g in (2,) + ((4,) if net.preferred_geometry == 4 else ()):   # both kernels where the shape has both
        net.default_geometry = g
        out = torch.zeros((B, M, co), device=dev)
        assert np.array_equal(net.grouped(X, F, NX, idx, out=out, cnt=cnt).cpu().numpy(), want), f"geometry {g}"
    net.default_geometry = 0
