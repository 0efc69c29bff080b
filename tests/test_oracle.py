"""CPU tests of the spec-oracle against INDEPENDENT restatements (numpy / scipy / torch) and the
committed golden vectors.  The reference ships no tests or fixtures (/root/reference/README.md:1-2),
so these cross-checks are what keeps the oracle from merely confirming itself."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _d2(p, c):
    """SPEC.md §1 in float32 with the same operation order (numpy float32 ops never fuse)."""
    d = (p - c).astype(np.float32)
    sq = (d * d).astype(np.float32)
    return ((sq[..., 0] + sq[..., 1]).astype(np.float32) + sq[..., 2]).astype(np.float32)


def _rand_xyz(seed, B, N, scale=1.0):
    return (np.random.default_rng(seed).uniform(0, 1, (B, N, 3)) * scale).astype(np.float32)


# ---------------------------------------------------------------- fps
def _fps_numpy(xyz, M):
    out = np.zeros((xyz.shape[0], M), np.int32)
    for b in range(xyz.shape[0]):
        mind = np.full(xyz.shape[1], np.inf, np.float32)
        last = 0
        for i in range(1, M):
            mind = np.minimum(mind, _d2(xyz[b], xyz[b, last]))
            last = int(np.argmax(mind))  # first occurrence = lowest index
            out[b, i] = last
    return out


@pytest.mark.parametrize("B,N,M", [(2, 257, 64), (1, 1024, 256), (3, 100, 100)])
def test_fps_vs_numpy(orc, B, N, M):
    xyz = _rand_xyz(1, B, N)
    got = orc.fps(xyz, M)
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, _fps_numpy(xyz, M))
    assert (got[:, 0] == 0).all()
    for b in range(B):
        assert len(set(got[b].tolist())) == M  # distinct while enough distinct points exist


def test_fps_duplicates_and_ties(orc):
    g = np.load(os.path.join(GOLDEN, "edge_cases.npz"))
    # collinear 0..9 from index 0: farthest is 9, then 4 (ties 4/5 -> distance 4 vs 4 -> lowest index)
    np.testing.assert_array_equal(orc.fps(g["line_xyz"], 5), g["line_fps5"])
    assert g["line_fps5"][0, :3].tolist() == [0, 9, 4]
    got = orc.fps(g["dup_xyz"], 6)
    np.testing.assert_array_equal(got, g["dup_fps6"])
    assert got[0, :3].tolist() == [0, 7, 6]      # (0,2,0) is farther than (1,0,0)
    assert got[0, 3:].tolist() == [0, 0, 0]      # every min-distance is 0 -> argmax returns 0


# ---------------------------------------------------------------- ball query
def _bq_numpy(r, S, xyz, new_xyz):
    B, M = new_xyz.shape[:2]
    out = np.zeros((B, M, S), np.int32)
    for b in range(B):
        for m in range(M):
            rr = np.float32(r[b, m]) if np.ndim(r) else np.float32(r)
            hit = np.nonzero(_d2(xyz[b], new_xyz[b, m]) < np.float32(rr * rr))[0]
            if len(hit):
                out[b, m, :] = hit[0]
                k = min(S, len(hit))
                out[b, m, :k] = hit[:k]
    return out


@pytest.mark.parametrize("r,S", [(0.1, 8), (0.2, 32), (0.45, 16), (2.0, 64)])
def test_ball_query_vs_numpy(orc, r, S):
    xyz = _rand_xyz(2, 2, 700)
    new_xyz = xyz[:, ::7].copy()
    np.testing.assert_array_equal(orc.ball_query(r, S, xyz, new_xyz), _bq_numpy(r, S, xyz, new_xyz))


def test_ball_query_sets_vs_ckdtree(orc):
    """Independent library: scipy's KD-tree (float64) must see the same in-radius SETS wherever no
    point sits within rounding distance of the sphere."""
    from scipy.spatial import cKDTree
    xyz = _rand_xyz(3, 1, 2000)
    new_xyz = xyz[:, :50].copy()
    r, S = 0.15, 64
    got = orc.ball_query(r, S, xyz, new_xyz)
    tree = cKDTree(xyz[0].astype(np.float64))
    for m in range(50):
        d = np.linalg.norm(xyz[0].astype(np.float64) - new_xyz[0, m].astype(np.float64), axis=1)
        if np.any(np.abs(d - np.float64(np.float32(r))) < 1e-6):
            continue
        want = sorted(tree.query_ball_point(new_xyz[0, m].astype(np.float64), float(np.float32(r))))
        assert sorted(set(got[0, m].tolist())) == want[:S] or len(want) > S
        if len(want) <= S:
            assert sorted(set(got[0, m].tolist())) == want


def test_ball_query_adaptive_equals_scalar_when_constant(orc):
    xyz = _rand_xyz(4, 2, 500)
    new_xyz = xyz[:, :40].copy()
    const = np.full((2, 40), 0.3, np.float32)
    np.testing.assert_array_equal(orc.ball_query(const, 16, xyz, new_xyz),
                                  orc.ball_query(0.3, 16, xyz, new_xyz))
    rad = np.random.default_rng(5).uniform(0.05, 0.6, (2, 40)).astype(np.float32)
    np.testing.assert_array_equal(orc.ball_query(rad, 16, xyz, new_xyz), _bq_numpy(rad, 16, xyz, new_xyz))


def test_ball_query_edge_cases(orc):
    g = np.load(os.path.join(GOLDEN, "edge_cases.npz"))
    got = orc.ball_query(2.0, 4, g["line_xyz"], g["line_cen"])
    np.testing.assert_array_equal(got, g["line_bq_r2_s4"])
    assert got[0, 0].tolist() == [3, 4, 5, 6]          # |x-4.5| < 2 -> 3..6 ; exactly nsample
    assert got[0, 1].tolist() == [0, 1, 0, 0]          # 0,1 accepted (2 is AT the radius: strict)
    assert got[0, 2].tolist() == [0, 0, 0, 0]          # empty ball -> zeros
    got = orc.ball_query(1.5, 8, g["line_xyz"], g["line_cen"])
    np.testing.assert_array_equal(got, g["line_bq_r1.5_s8"])
    assert got[0, 0].tolist() == [4, 5, 4, 4, 4, 4, 4, 4]   # padded with the FIRST accepted index
    got = orc.ball_query(5.0, 4, g["strict_xyz"], g["strict_xyz"][:, :1])
    np.testing.assert_array_equal(got, g["strict_bq"])
    assert got[0, 0].tolist() == [0, 3, 0, 0]          # (3,4,0) and (0,5,0) are exactly at r=5
    np.testing.assert_array_equal(orc.ball_query(0.5, 4, g["dup_xyz"], g["dup_xyz"][:, :3]), g["dup_bq"])


# ---------------------------------------------------------------- knn
def test_knn_vs_lexsort(orc):
    xyz = _rand_xyz(6, 2, 300)
    xyz[0, 10] = xyz[0, 20]  # a duplicate -> exact tie in d2
    new_xyz = xyz[:, :30].copy()
    got = orc.knn_query(9, xyz, new_xyz)
    for b in range(2):
        for m in range(30):
            d = _d2(xyz[b], new_xyz[b, m])
            want = np.lexsort((np.arange(300), d))[:9]
            np.testing.assert_array_equal(got[b, m], want)
    g = np.load(os.path.join(GOLDEN, "edge_cases.npz"))
    got = orc.knn_query(3, g["line_xyz"], g["line_cen"])
    np.testing.assert_array_equal(got, g["line_knn3"])
    assert got[0, 0].tolist() == [4, 5, 3]             # 4 and 5 tie -> lower index first


def test_knn_vs_torch_topk(orc):
    import torch
    xyz = _rand_xyz(7, 1, 512)
    new_xyz = xyz[:, :64].copy()
    got = orc.knn_query(8, xyz, new_xyz)
    d = torch.cdist(torch.from_numpy(new_xyz).double(), torch.from_numpy(xyz).double())
    want = torch.topk(d, 8, dim=2, largest=False).indices.numpy()
    assert (np.sort(got, 2) == np.sort(want, 2)).mean() > 0.999  # float64 vs float32 near-ties only


# ---------------------------------------------------------------- group / gather
def test_group_and_gather_vs_numpy(orc):
    rng = np.random.default_rng(8)
    feat = rng.normal(size=(2, 5, 77)).astype(np.float32)
    idx = rng.integers(0, 77, (2, 9, 4)).astype(np.int32)
    want = np.stack([feat[b][:, idx[b]] for b in range(2)])
    np.testing.assert_array_equal(orc.group_points(feat, idx), want)
    i2 = idx[:, :, 0].copy()
    np.testing.assert_array_equal(orc.gather_points(feat, i2), np.stack([feat[b][:, i2[b]] for b in range(2)]))
    h = feat.astype(np.float16)
    np.testing.assert_array_equal(orc.group_points(h, idx), np.stack([h[b][:, idx[b]] for b in range(2)]))
    xyz = rng.normal(size=(2, 77, 3)).astype(np.float32)
    np.testing.assert_array_equal(orc.gather_xyz(xyz, i2), np.stack([xyz[b][i2[b]] for b in range(2)]))


# ---------------------------------------------------------------- MLP
def _fma_chain(W, b, x, relu):
    """Pure-Python/float64-exact emulation of the binary32 fmaf chain of SPEC.md §6:
    a*b is exact in float64 (24+24 bits), acc+prod in float64 then ONE rounding to float32 is the
    fmaf result except in double-rounding corner cases, which math.fma-free Python cannot avoid;
    so this check uses a 1-ulp tolerance while torch/float64 checks bound the overall error."""
    out = np.zeros((x.shape[0], W.shape[0]), np.float32)
    for r in range(x.shape[0]):
        for o in range(W.shape[0]):
            acc = np.float32(b[o])
            for k in range(W.shape[1]):
                acc = np.float32(np.float64(W[o, k]) * np.float64(x[r, k]) + np.float64(acc))
            out[r, o] = acc if (acc > 0 or not relu) else np.float32(0)
    return out


def test_mlp_rows_vs_chain_and_float64(orc):
    import sad_amd  # noqa: F401
    from sad_amd import synth
    rng = np.random.default_rng(9)
    layers = synth.make_mlp_weights([7, 12, 5], rng)
    x = rng.normal(size=(37, 7)).astype(np.float32)
    got = orc.mlp_rows(x, layers)
    h = _fma_chain(layers[0][0], layers[0][1], x, True)
    want = _fma_chain(layers[1][0], layers[1][1], h, True)
    np.testing.assert_allclose(got, want, rtol=3e-7, atol=1e-7)
    ref = x.astype(np.float64)
    for W, b in layers:
        ref = np.maximum(ref @ W.astype(np.float64).T + b.astype(np.float64), 0)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    # relu_mask: last layer linear
    got2 = orc.mlp_rows(x, layers, relu_mask=1)
    ref2 = np.maximum(x.astype(np.float64) @ layers[0][0].astype(np.float64).T + layers[0][1], 0)
    ref2 = ref2 @ layers[1][0].astype(np.float64).T + layers[1][1]
    np.testing.assert_allclose(got2, ref2, rtol=1e-5, atol=1e-5)
    assert (got2 < 0).any()


def test_mlp_vs_torch_conv2d(orc):
    """Independent library: torch.nn.functional.conv2d (1x1) on the grouped tensor."""
    import torch
    import torch.nn.functional as F
    import sad_amd  # noqa: F401
    from sad_amd import synth
    rng = np.random.default_rng(10)
    xyz = _rand_xyz(11, 2, 200)
    feat_cm = rng.normal(size=(2, 6, 200)).astype(np.float32)          # [B,C,N]
    fidx = orc.fps(xyz, 20)
    new_xyz = orc.gather_xyz(xyz, fidx)
    idx = orc.ball_query(0.3, 8, xyz, new_xyz)
    layers = synth.make_mlp_weights([9, 16, 24], rng)
    got = orc.sa_group_mlp_max(xyz, np.ascontiguousarray(feat_cm.transpose(0, 2, 1)), new_xyz, idx, layers)
    gx = orc.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    g = torch.from_numpy(np.concatenate([gx, orc.group_points(feat_cm, idx)], 1))   # [B,9,M,S]
    for W, b in layers:
        g = F.relu(F.conv2d(g, torch.from_numpy(W)[:, :, None, None], torch.from_numpy(b)))
    want = g.max(dim=3).values.permute(0, 2, 1).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_fused_equals_composition(orc):
    """sa_group_mlp_max == group -> mlp_rows -> max, bit for bit (same fmaf chains)."""
    import sad_amd  # noqa: F401
    from sad_amd import synth
    rng = np.random.default_rng(12)
    xyz = _rand_xyz(13, 1, 300)
    feat = rng.normal(size=(1, 300, 4)).astype(np.float32)
    fidx = orc.fps(xyz, 16)
    new_xyz = orc.gather_xyz(xyz, fidx)
    idx = orc.ball_query(0.4, 16, xyz, new_xyz)
    layers = synth.make_mlp_weights([7, 8, 8, 16], rng)
    fused = orc.sa_group_mlp_max(xyz, feat, new_xyz, idx, layers)
    rows = np.concatenate([xyz[0][idx[0]] - new_xyz[0][:, None, :], feat[0][idx[0]]], -1).reshape(-1, 7)
    y = orc.mlp_rows(rows.astype(np.float32), layers).reshape(16, 16, 16).max(1)
    np.testing.assert_array_equal(fused[0], y)
    # writing into a slice of a wider buffer (branch concatenation)
    buf = np.full((1, 16, 40), -7, np.float32)
    orc.sa_group_mlp_max(xyz, feat, new_xyz, idx, layers, out=buf, col_off=8)
    np.testing.assert_array_equal(buf[0, :, 8:24], y)
    assert (buf[0, :, :8] == -7).all() and (buf[0, :, 24:] == -7).all()


# ---------------------------------------------------------------- cluster layer + head
def test_candidates_and_decode_vs_numpy(orc):
    import sad_amd  # noqa: F401
    from sad_amd import config
    rng = np.random.default_rng(14)
    xyz3 = rng.uniform(-5, 5, (2, 20, 3)).astype(np.float32)
    c = rng.normal(0, 1.5, (2, 8, 6)).astype(np.float32)
    cand, rad = orc.candidates(xyz3, c, config.SHIFT_MAX, config.R_MIN, config.R_MAX, config.ANCHOR_CAR)
    sh = np.clip(c[..., :3], -2, 2)
    np.testing.assert_array_equal(cand, xyz3[:, :8] + sh)
    s = np.clip(c[..., 3:], -1, 1).astype(np.float32)
    q = ((np.float32(1) + s) + (np.float32(0.5) * (s * s))).astype(np.float32)
    sz = (np.asarray(config.ANCHOR_CAR, np.float32) * q).astype(np.float32)
    sq = (sz * sz).astype(np.float32)
    r = np.float32(0.5) * np.sqrt(((sq[..., 0] + sq[..., 1]).astype(np.float32) + sq[..., 2]).astype(np.float32))
    np.testing.assert_array_equal(rad, np.clip(r, config.R_MIN, config.R_MAX).astype(np.float32))
    assert rad.min() >= config.R_MIN and rad.max() <= config.R_MAX and rad.std() > 0
    o = rng.normal(0, 1, (2, 8, 10)).astype(np.float32)
    boxes = orc.decode_boxes(cand, o, config.ANCHORS)
    lab = np.argmax(o[..., :3], -1)
    np.testing.assert_array_equal(boxes[..., 8], lab.astype(np.float32))
    np.testing.assert_allclose(boxes[..., 7], 1 / (1 + np.exp(-np.max(o[..., :3], -1))), rtol=1e-6)
    np.testing.assert_array_equal(boxes[..., :3], cand + o[..., 3:6])
    anc = np.asarray(config.ANCHORS, np.float32)[lab]
    np.testing.assert_allclose(boxes[..., 3:6], anc * np.exp(np.clip(o[..., 6:9], -2, 2)), rtol=1e-6)
    np.testing.assert_array_equal(boxes[..., 6], o[..., 9])


# ---------------------------------------------------------------- golden vectors
def test_golden_config0(orc):
    """BASELINE.json configs[0] (1x1024 pts, npoint=256, r=0.2, nsample=32) pinned."""
    import sad_amd  # noqa: F401
    from sad_amd import config, synth
    g = np.load(os.path.join(GOLDEN, "config0.npz"))
    np.testing.assert_array_equal(synth.make_unit_cube(0, 1024)[None], g["xyz"])  # generator is stable
    st = config.CONFIG0_SA
    fidx = orc.fps(g["xyz"], st.npoint)
    np.testing.assert_array_equal(fidx, g["fps_idx"])
    new_xyz = orc.gather_xyz(g["xyz"], fidx)
    np.testing.assert_array_equal(new_xyz, g["new_xyz"])
    bidx = orc.ball_query(st.radii[0], st.nsamples[0], g["xyz"], new_xyz)
    np.testing.assert_array_equal(bidx, g["ball_idx"])
    np.testing.assert_array_equal(orc.knn_query(16, g["xyz"], new_xyz), g["knn_idx"])
    layers = synth.make_mlp_weights([3, 64, 64, 128], np.random.default_rng(7))
    np.testing.assert_array_equal(orc.sa_group_mlp_max(g["xyz"], None, new_xyz, bidx, layers), g["feat"])


def test_golden_adaptive(orc):
    import sad_amd  # noqa: F401
    from sad_amd import synth
    g = np.load(os.path.join(GOLDEN, "adaptive.npz"))
    xyz = np.ascontiguousarray(synth.make_tiny_batch(100, 2, 2048)[:, :, :3])
    fidx = orc.fps(xyz, 128)
    np.testing.assert_array_equal(fidx, g["fps_idx"])
    np.testing.assert_array_equal(orc.ball_query(g["radius"], 16, xyz, orc.gather_xyz(xyz, fidx)), g["ball_idx"])


def test_golden_tiny_detector(orc):
    import sad_amd  # noqa: F401
    from sad_amd import config, synth
    g = np.load(os.path.join(GOLDEN, "tiny_detector.npz"))
    cfg = config.TINY
    tr = {}
    boxes = orc.detector_forward(synth.make_tiny_batch(0, 2, cfg.n_points), cfg, synth.make_weights(cfg, 0), tr)
    for k in ("sa1", "sa2", "sa3"):
        np.testing.assert_array_equal(tr[k]["fps_idx"], g[f"{k}_fps"])
    np.testing.assert_array_equal(tr["sa3"]["out"], g["sa3_out"])
    np.testing.assert_array_equal(tr["cluster"]["radius"], g["radius"])
    np.testing.assert_array_equal(tr["cluster"]["ball_idx"][0], g["cl_idx0"])
    np.testing.assert_array_equal(tr["cluster"]["ball_idx"][1], g["cl_idx1"])
    np.testing.assert_allclose(boxes, g["boxes"], rtol=1e-6, atol=1e-6)  # expf may differ per libm
    assert g["radius"].std() > 0.01  # the adaptive radius really varies per candidate


# ---------------------------------------------------------------- rotated NMS (SPEC.md §13)
def _random_boxes(seed, B, K, extent=30.0):
    rng = np.random.default_rng(seed)
    bx = np.zeros((B, K, 9), np.float32)
    bx[..., 0:2] = rng.uniform(0, extent, (B, K, 2))
    bx[..., 2] = rng.uniform(-2, 0, (B, K))
    bx[..., 3] = rng.uniform(2.5, 5.0, (B, K))
    bx[..., 4] = rng.uniform(1.2, 2.2, (B, K))
    bx[..., 5] = rng.uniform(1.2, 2.0, (B, K))
    bx[..., 6] = rng.uniform(-7, 7, (B, K))
    bx[..., 7] = rng.uniform(0, 1, (B, K))
    bx[..., 8] = rng.integers(0, 3, (B, K))
    return bx


def _iou_float64(a, b):
    """Independent restatement: float64 corners from numpy sin/cos + Sutherland-Hodgman."""
    def corners(q):
        c, s = np.cos(np.float64(q[6])), np.sin(np.float64(q[6]))
        d = np.array([[1, 1], [-1, 1], [-1, -1], [1, -1]], np.float64) * np.array([q[3], q[4]], np.float64) / 2
        return np.stack([q[0] + c * d[:, 0] - s * d[:, 1], q[1] + s * d[:, 0] + c * d[:, 1]], 1)
    poly, clip = corners(a), corners(b)
    for e in range(4):
        q0, q1 = clip[e], clip[(e + 1) % 4]
        out = []
        for i in range(len(poly)):
            cur, prev = poly[i], poly[i - 1]
            cc = (q1[0] - q0[0]) * (cur[1] - q0[1]) - (q1[1] - q0[1]) * (cur[0] - q0[0])
            cp = (q1[0] - q0[0]) * (prev[1] - q0[1]) - (q1[1] - q0[1]) * (prev[0] - q0[0])
            if (cc >= 0) != (cp >= 0):
                out.append(prev + cp / (cp - cc) * (cur - prev))
            if cc >= 0:
                out.append(cur)
        poly = np.array(out) if out else np.zeros((0, 2))
        if len(poly) == 0:
            break
    inter = 0.0
    if len(poly) >= 3:
        x, y = poly[:, 0], poly[:, 1]
        inter = 0.5 * abs(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))
    return inter / (a[3] * a[4] + b[3] * b[4] - inter)


def test_sincos_r_accuracy(orc):
    th = np.linspace(-50, 50, 20001).astype(np.float32)
    s, c = orc.sincos_r(th)
    assert np.abs(s - np.sin(th.astype(np.float64))).max() < 3e-7
    assert np.abs(c - np.cos(th.astype(np.float64))).max() < 3e-7


def test_iou_bev_vs_float64_and_known_answers(orc):
    bx = _random_boxes(1, 1, 400, extent=12.0)[0]
    a, b = bx[:200], bx[200:]
    got = orc.iou_bev(a, b)
    want = np.array([_iou_float64(p, q) for p, q in zip(a, b)])
    assert (want > 0.05).sum() >= 15                      # the sample really contains overlaps
    np.testing.assert_allclose(got, want, atol=2e-5)
    unit = np.array([[0, 0, 0, 4, 2, 1, 0, 1, 0]], np.float32)
    assert orc.iou_bev(unit, unit)[0] == 1.0
    assert orc.iou_bev(unit, unit + np.array([10, 0, 0, 0, 0, 0, 0, 0, 0], np.float32))[0] == 0.0
    np.testing.assert_allclose(orc.iou_bev(unit, unit + np.array([2, 0, 0, 0, 0, 0, 0, 0, 0], np.float32)), [1 / 3], rtol=1e-6)
    rot = unit.copy()
    rot[0, 6] = np.pi / 2
    np.testing.assert_allclose(orc.iou_bev(unit, rot), [4 / 12], rtol=1e-5)


def test_nms_bev_vs_python(orc):
    bx = _random_boxes(2, 3, 120, extent=25.0)
    bx[1, 5, 7] = bx[1, 9, 7]                               # exact score tie -> lower index first
    keep, order, count = orc.nms_bev(bx, 0.1, 0.2)
    for b in range(3):
        cand = [i for i in range(120) if bx[b, i, 7] >= np.float32(0.2)]
        cand.sort(key=lambda i: (-bx[b, i, 7], i))
        kept = []
        for i in cand:
            if all(orc.iou_bev(bx[b, k], bx[b, i])[0] <= np.float32(0.1) for k in kept):
                kept.append(i)
        assert count[b] == len(kept) and order[b, :len(kept)].tolist() == kept
        assert (order[b, len(kept):] == -1).all()
        assert keep[b].sum() == len(kept) and all(keep[b, i] == 1 for i in kept)
        assert 0 < len(kept) < len(cand)                    # something was suppressed


def test_bf16_oracle_rounding_and_chain(orc):
    """SPEC.md §14: bf16_round is round-to-nearest-even (checked against torch's conversion), and the
    bf16 chain equals a float64 matmul of the rounded operands."""
    import torch
    rng = np.random.default_rng(14)
    x = (rng.standard_normal(20000) * np.exp(rng.uniform(-20, 20, 20000))).astype(np.float32)
    x[:4] = [0.0, -0.0, 1.00390625, 1.01171875]          # ties: 1 + 2^-8 -> even (1.0), 1 + 3*2^-8 -> 1.015625
    got = orc.bf16_round(x)
    want = torch.from_numpy(x).bfloat16().float().numpy()
    assert np.array_equal(got, want)
    assert got[2] == 1.0 and got[3] == 1.015625
    W = rng.standard_normal((5, 7)).astype(np.float32)
    b = rng.standard_normal(5).astype(np.float32)
    rows = rng.standard_normal((11, 7)).astype(np.float32)
    y = orc.mlp_rows_bf16(rows, [(W, b)], relu_mask=0)
    ref = orc.bf16_round(rows).astype(np.float64) @ orc.bf16_round(W).astype(np.float64).T + b
    assert np.allclose(y, ref, rtol=1e-6, atol=1e-6)


def test_ffps_oracle_reduces_to_fps_and_is_greedy(orc):
    """SPEC.md §15: zero features with w_xyz = 1 give plain FPS; with features every pick maximises
    the min metric distance to the earlier picks (recomputed in numpy, same evaluation order)."""
    rng = np.random.default_rng(151)
    xyz = rng.random((2, 200, 3), dtype=np.float32)
    np.testing.assert_array_equal(orc.ffps(xyz, np.zeros((2, 200, 3), np.float32), 50), orc.fps(xyz, 50))
    feat = rng.standard_normal((2, 200, 4)).astype(np.float32)
    idx = orc.ffps(xyz, feat, 30, 0.5)

    def metric(b, q):
        d = xyz[b] - xyz[b][q]
        d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]) * np.float32(0.5)
        for c in range(4):
            t = feat[b][:, c] - feat[b][q, c]
            d2 = d2 + t * t
        return d2
    for b in range(2):
        mind = np.full(200, np.inf, np.float32)
        for i in range(1, 30):
            mind = np.minimum(mind, metric(b, idx[b, i - 1]))
            assert int(np.argmax(mind)) == idx[b, i]


def test_golden_extensions(orc):
    """SPEC §13-§15 operators against the committed vectors (pins the oracle against drift)."""
    g = np.load(os.path.join(GOLDEN, "extensions.npz"))
    np.testing.assert_array_equal(orc.ffps(g["ffps_xyz"], g["ffps_feat"], 64, 1.0), g["ffps_idx_w1"])
    np.testing.assert_array_equal(orc.ffps(g["ffps_xyz"], g["ffps_feat"], 64, 0.0), g["ffps_idx_w0"])
    layers = [(g[f"bf16_w{i}"], g[f"bf16_b{i}"]) for i in range(2)]
    xyz, feat = g["ffps_xyz"], g["ffps_feat"]
    new_xyz = np.ascontiguousarray(xyz[:, :50])
    np.testing.assert_array_equal(orc.ball_query(0.2, 16, xyz, new_xyz), g["bf16_idx"])
    np.testing.assert_array_equal(orc.sa_group_mlp_max_bf16(xyz, orc.bf16_round(feat), new_xyz, g["bf16_idx"], layers),
                                  g["bf16_pooled"])
    np.testing.assert_array_equal(orc.mlp_rows_bf16(np.concatenate([xyz[0, :32], feat[0, :32]], 1), layers), g["bf16_rows"])
    keep, order, count = orc.nms_bev(g["nms_boxes"], 0.1, 0.2)
    np.testing.assert_array_equal(keep, g["nms_keep"])
    np.testing.assert_array_equal(order, g["nms_order"])
    np.testing.assert_array_equal(count, g["nms_count"])
