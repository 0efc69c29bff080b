"""ctypes binding of the CPU spec-oracle (``oracle/sad_oracle.c``) + the CPU model forward.

TEST INFRASTRUCTURE ONLY: importable from ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product package (``3dsad-main_amd/``) never imports it.

PARITY UNPINNED: the upstream reference (``/root/reference/README.md:1-2``) contains no code, tests
or golden vectors, so this oracle restates this repository's SPEC.md, not a reference file.
All functions take and return numpy arrays (float32 / int32, C-contiguous).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAD_ORACLE_LIB: an alternative build of the same oracle (the -fsanitize=address,undefined one,
# oracle/Makefile target `asan`, loaded by tests/test_oracle_sanitizers.py in a child process)
_SO = os.environ.get("SAD_ORACLE_LIB") or os.path.join(_HERE, "libsad_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Returns the path of the shared library."""
    src = os.path.join(_HERE, "sad_oracle.c")
    if os.environ.get("SAD_ORACLE_LIB"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsad_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def fps(xyz, npoint):
    """SPEC.md §2.  xyz [B,N,3] -> idx [B,npoint] int32."""
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    assert 1 <= npoint <= N
    idx = np.empty((B, npoint), np.int32)
    lib().orc_fps(px, B, N, npoint, idx.ctypes.data_as(_i32p))
    return idx


def ffps(xyz, feat_pm, npoint, w_xyz=1.0):
    """SPEC.md §15.  xyz [B,N,3], feat_pm [B,N,C] -> idx [B,npoint] (feature-distance FPS)."""
    xyz, px = _f(xyz)
    feat_pm, pf = _f(feat_pm)
    B, N, _ = xyz.shape
    C = feat_pm.shape[2]
    idx = np.empty((B, npoint), np.int32)
    lib().orc_ffps(px, pf, B, N, C, npoint, ctypes.c_float(w_xyz), idx.ctypes.data_as(_i32p))
    return idx


def ball_query(radius, nsample, xyz, new_xyz):
    """SPEC.md §3.  radius: python float or [B,M] array (adaptive).  -> idx [B,M,nsample]."""
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.empty((B, M, nsample), np.int32)
    if np.ndim(radius) == 0:
        lib().orc_ball_query(px, pn, B, N, M, nsample, ctypes.c_float(float(np.float32(radius))),
                             None, idx.ctypes.data_as(_i32p))
    else:
        rad, pr = _f(radius)
        assert rad.shape == (B, M)
        lib().orc_ball_query(px, pn, B, N, M, nsample, ctypes.c_float(0.0), pr,
                             idx.ctypes.data_as(_i32p))
    return idx


def knn_query(k, xyz, new_xyz):
    """SPEC.md §4.  -> idx [B,M,k] sorted by (d2, index)."""
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    assert k <= 64 and k <= N
    idx = np.empty((B, M, k), np.int32)
    lib().orc_knn(px, pn, B, N, M, k, idx.ctypes.data_as(_i32p))
    return idx


def gather_points(src, idx):
    """SPEC.md §5.  src [B,C,N] (2- or 4-byte dtype), idx [B,M] -> [B,C,M]."""
    src = np.ascontiguousarray(src)
    idx, pi = _i(idx)
    B, C, N = src.shape
    M = idx.shape[1]
    out = np.empty((B, C, M), src.dtype)
    lib().orc_gather_points(ctypes.c_void_p(src.ctypes.data), pi, B, C, N, M, src.itemsize,
                            ctypes.c_void_p(out.ctypes.data))
    return out


def gather_xyz(xyz, idx):
    xyz, px = _f(xyz)
    idx, pi = _i(idx)
    B, N, _ = xyz.shape
    M = idx.shape[1]
    out = np.empty((B, M, 3), np.float32)
    lib().orc_gather_xyz(px, pi, B, N, M, out.ctypes.data_as(_f32p))
    return out


def group_points(feat, idx):
    """SPEC.md §5.  feat [B,C,N], idx [B,M,S] -> [B,C,M,S]."""
    feat = np.ascontiguousarray(feat)
    idx, pi = _i(idx)
    B, C, N = feat.shape
    _, M, S = idx.shape
    out = np.empty((B, C, M, S), feat.dtype)
    lib().orc_group_points(ctypes.c_void_p(feat.ctypes.data), pi, B, C, N, M, S, feat.itemsize,
                           ctypes.c_void_p(out.ctypes.data))
    return out


def _wb(layers):
    L = len(layers)
    Ws = [np.ascontiguousarray(w, np.float32) for w, _ in layers]
    bs = [np.ascontiguousarray(b, np.float32) for _, b in layers]
    dims = (ctypes.c_int * (L + 1))(*([Ws[0].shape[1]] + [w.shape[0] for w in Ws]))
    for a, b in zip(Ws[:-1], Ws[1:]):
        assert b.shape[1] == a.shape[0]
    Wp = (_f32p * L)(*[w.ctypes.data_as(_f32p) for w in Ws])
    bp = (_f32p * L)(*[b.ctypes.data_as(_f32p) for b in bs])
    return L, dims, Wp, bp, (Ws, bs)


def mlp_rows(x, layers, relu_mask=None, out=None, col_off=0):
    """SPEC.md §6 on plain rows.  x [R,C_in]; layers [(W,b),...]; relu_mask bit l = ReLU after l
    (default: every layer).  Returns [R,C_out] or writes into ``out[:, col_off:col_off+C_out]``."""
    x, px = _f(x)
    R = x.shape[0]
    L, dims, Wp, bp, keep = _wb(layers)
    assert x.shape[1] == dims[0]
    if relu_mask is None:
        relu_mask = (1 << L) - 1
    cout = dims[L]
    if out is None:
        out = np.empty((R, cout), np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape[0] == R
    lib().orc_mlp_rows(px, ctypes.c_int64(R), L, dims, Wp, bp, int(relu_mask),
                       out.ctypes.data_as(_f32p), out.shape[1], col_off)
    return out


def sa_group_mlp_max(xyz, feat_pm, new_xyz, idx, layers, out=None, col_off=0, skip_padding=False):
    """SPEC.md §6 fused.  xyz [B,N,3]; feat_pm [B,N,C] point-major or None; new_xyz [B,M,3];
    idx [B,M,S]; -> out [B,M,C_out] (or writes at channel offset col_off of a wider buffer).
    ``skip_padding``: do not compute the trailing rows of a group that repeat its first sample
    (same bits — a duplicate row cannot change a max; used by bench.py's like-for-like CPU leg)."""
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    idx, pi = _i(idx)
    B, N, _ = xyz.shape
    _, M, S = idx.shape
    if feat_pm is None:
        C, pf = 0, None
    else:
        feat_pm, pf = _f(feat_pm)
        C = feat_pm.shape[2]
        assert feat_pm.shape[:2] == (B, N)
    L, dims, Wp, bp, keep = _wb(layers)
    assert dims[0] == 3 + C, (dims[0], C)
    cout = dims[L]
    if out is None:
        out = np.empty((B, M, cout), np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape[:2] == (B, M)
    fn = lib().orc_sa_group_mlp_max_skip if skip_padding else lib().orc_sa_group_mlp_max
    fn(px, pf, pn, pi, B, N, M, S, C, L, dims, Wp, bp, out.ctypes.data_as(_f32p), out.shape[2], col_off)
    return out


def subsample_pad(points, offsets, n_points, seed=0):
    """SPEC.md §17.  points [total, C] f32, offsets [B+1] int32 -> out [B, n_points, C]."""
    points, pp = _f(points)
    offsets, po = _i(offsets)
    B = offsets.shape[0] - 1
    C = points.shape[1]
    out = np.empty((B, n_points, C), np.float32)
    lib().orc_subsample_pad(pp, po, B, C, int(n_points), ctypes.c_uint32(int(seed) & 0xFFFFFFFF), out.ctypes.data_as(_f32p))
    return out


def bf16_round(x):
    """SPEC.md §14: round-to-nearest-even binary32 -> bfloat16, returned as binary32 values."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def mlp_rows_bf16(x, layers, relu_mask=None):
    """SPEC.md §14 on rows x[R, C_in] (binary32 values, rounded to bf16 here).  layers: [(W, b)].
    Products of bf16 values are exact in binary64; the sum is rounded once to binary32."""
    L = len(layers)
    if relu_mask is None:
        relu_mask = (1 << L) - 1
    y = np.asarray(x, dtype=np.float32)
    for l, (W, b) in enumerate(layers):
        xb = bf16_round(y).astype(np.float64)
        Wb = bf16_round(np.asarray(W, dtype=np.float32)).astype(np.float64)
        y = (xb @ Wb.T + np.asarray(b, dtype=np.float64)).astype(np.float32)
        if (relu_mask >> l) & 1:
            y = np.maximum(y, 0.0)
    return y


def sa_group_mlp_max_bf16(xyz, feat_pm, new_xyz, idx, layers):
    """SPEC.md §14 grouped chain: xyz[B,N,3] f32, feat_pm[B,N,C] (bf16-representable f32) or None,
    new_xyz[B,M,3], idx[B,M,S] -> pooled [B,M,C_out] f32."""
    B, M, S = idx.shape
    out = np.empty((B, M, layers[-1][0].shape[0]), np.float32)
    for b in range(B):
        j = idx[b].reshape(-1)
        rel = (xyz[b][j] - np.repeat(new_xyz[b], S, axis=0)).astype(np.float32)
        rows = rel if feat_pm is None else np.concatenate([rel, feat_pm[b][j]], axis=1)
        y = mlp_rows_bf16(rows, layers)
        out[b] = y.reshape(M, S, -1).max(axis=1)
    return out


def candidates(xyz3, c, shift_max, r_min, r_max, anchor):
    """SPEC.md §8 steps 2-4.  xyz3 [B,M3,3], c [B,K,6] -> (cand [B,K,3], radius [B,K])."""
    xyz3, px = _f(xyz3)
    c, pc = _f(c)
    B, M3, _ = xyz3.shape
    K = c.shape[1]
    anchor, pa = _f(np.asarray(anchor, np.float32))
    cand = np.empty((B, K, 3), np.float32)
    rad = np.empty((B, K), np.float32)
    lib().orc_candidates(px, pc, B, M3, K, ctypes.c_float(shift_max), ctypes.c_float(r_min),
                         ctypes.c_float(r_max), pa, cand.ctypes.data_as(_f32p),
                         rad.ctypes.data_as(_f32p))
    return cand, rad


def decode_boxes(cand, o, anchors):
    """SPEC.md §9.  cand [B,K,3], o [B,K,10] -> boxes [B,K,9]."""
    cand, pc = _f(cand)
    o, po = _f(o)
    B, K, _ = cand.shape
    anchors, pa = _f(np.asarray(anchors, np.float32))
    boxes = np.empty((B, K, 9), np.float32)
    lib().orc_decode_boxes(pc, po, B, K, pa, boxes.ctypes.data_as(_f32p))
    return boxes


def sincos_r(theta):
    """SPEC.md §13 reproducible sin/cos."""
    th, pt = _f(np.ravel(theta))
    s = np.empty_like(th)
    c = np.empty_like(th)
    lib().orc_sincos_r(pt, th.size, s.ctypes.data_as(_f32p), c.ctypes.data_as(_f32p))
    return s.reshape(np.shape(theta)), c.reshape(np.shape(theta))


def iou_bev(a, b):
    """SPEC.md §13: rotated BEV IoU of box pairs a[i], b[i] (rows of 9 floats)."""
    a, pa = _f(np.reshape(a, (-1, 9)))
    b, pb = _f(np.reshape(b, (-1, 9)))
    out = np.empty((a.shape[0],), np.float32)
    lib().orc_iou_bev(pa, pb, a.shape[0], out.ctypes.data_as(_f32p))
    return out


def nms_bev(boxes, iou_thr, score_thr=0.0):
    """SPEC.md §13.  boxes [B,K,9] -> (keep [B,K] int32, order [B,K] int32 (-1 padded), count [B])."""
    boxes, pb = _f(boxes)
    B, K, _ = boxes.shape
    keep = np.empty((B, K), np.int32)
    order = np.empty((B, K), np.int32)
    count = np.empty((B,), np.int32)
    lib().orc_nms_bev(pb, B, K, ctypes.c_float(float(np.float32(iou_thr))),
                      ctypes.c_float(float(np.float32(score_thr))), keep.ctypes.data_as(_i32p),
                      order.ctypes.data_as(_i32p), count.ctypes.data_as(_i32p))
    return keep, order, count


# ----------------------------------------------------------------------------------------------
# Model-level restatement (SPEC.md §7-§9): the CPU path the GPU detector is compared with.
# ----------------------------------------------------------------------------------------------
def sa_module(xyz, feat_pm, stage, weights, name, trace=None, skip_padding=False):
    """SPEC.md §7 with point-major features.  xyz [B,N,3], feat_pm [B,N,C] or None.
    weights[name+'.b<i>'] per branch, weights[name+'.agg'] if stage.agg.
    Returns (new_xyz [B,M,3], new_feat_pm [B,M,C'])."""
    B = xyz.shape[0]
    M = stage.npoint
    fidx = fps(xyz, M)
    new_xyz = gather_xyz(xyz, fidx)
    cat = sum(m[-1] for m in stage.mlps)
    buf = np.empty((B, M, cat), np.float32)
    off = 0
    idxs = []
    for bi, (r, s, mlp) in enumerate(zip(stage.radii, stage.nsamples, stage.mlps)):
        idx = ball_query(r, s, xyz, new_xyz)
        idxs.append(idx)
        sa_group_mlp_max(xyz, feat_pm, new_xyz, idx, weights[f"{name}.b{bi}"], out=buf, col_off=off,
                         skip_padding=skip_padding)
        off += mlp[-1]
    if stage.agg:
        out = mlp_rows(buf.reshape(B * M, cat), weights[f"{name}.agg"]).reshape(B, M, stage.agg)
    else:
        out = buf
    if trace is not None:
        trace[name] = dict(fps_idx=fidx, new_xyz=new_xyz, ball_idx=idxs, cat=buf, out=out)
    return new_xyz, out


def detector_forward(points, cfg, weights, trace=None, skip_padding=False):
    """SPEC.md §7-§9: points [B,N,3+in_feat] -> boxes [B,K,9].  ``skip_padding``: see
    ``sa_group_mlp_max`` (identical boxes, fewer MLP rows)."""
    points = np.ascontiguousarray(points, np.float32)
    B = points.shape[0]
    xyz = np.ascontiguousarray(points[:, :, :3])
    feat = np.ascontiguousarray(points[:, :, 3:]) if points.shape[2] > 3 else None
    for si, st in enumerate(cfg.stages):
        xyz, feat = sa_module(xyz, feat, st, weights, f"sa{si + 1}", trace, skip_padding)
    K = cfg.n_cand
    C3 = feat.shape[2]
    cand_feat = np.ascontiguousarray(feat[:, :K, :]).reshape(B * K, C3)
    L = len(weights["cand"])
    c = mlp_rows(cand_feat, weights["cand"], relu_mask=(1 << (L - 1)) - 1).reshape(B, K, 6)
    cand, rad = candidates(xyz, c, cfg.shift_max, cfg.r_min, cfg.r_max, cfg.anchor_car)
    cat = sum(m[-1] for m in cfg.cluster_mlps)
    buf = np.empty((B, K, cat), np.float32)
    off = 0
    cidx = []
    for bi, (sc, s, mlp) in enumerate(zip(cfg.cluster_scales, cfg.cluster_nsamples, cfg.cluster_mlps)):
        r = (np.float32(sc) * rad).astype(np.float32)
        idx = ball_query(r, s, xyz, cand)
        cidx.append(idx)
        sa_group_mlp_max(xyz, feat, cand, idx, weights[f"cluster.b{bi}"], out=buf, col_off=off,
                         skip_padding=skip_padding)
        off += mlp[-1]
    cfeat = mlp_rows(buf.reshape(B * K, cat), weights["cluster.agg"])
    Lh = len(weights["head"])
    o = mlp_rows(cfeat, weights["head"], relu_mask=(1 << (Lh - 1)) - 1).reshape(B, K, 10)
    boxes = decode_boxes(cand, o, cfg.anchors)
    if trace is not None:
        trace["cluster"] = dict(c=c, cand=cand, radius=rad, ball_idx=cidx, cat=buf, cfeat=cfeat,
                                head=o, xyz3=xyz, feat3=feat)
    return boxes
