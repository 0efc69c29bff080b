"""Python operator surface: ``fps / ball_query / knn_query / group_points / gather_points`` plus
the fused grouped-MLP (``PackedMLP`` / ``mlp_chain``).

Names and argument order are the ones BASELINE.json ``north_star`` fixes ("keeps the reference's
Python operator surface (fps / ball_query / group_points / sa_module)"); the upstream reference
itself (``/root/reference/README.md:1-2``) defines none, so semantics are SPEC.md §2-§6.

Every function takes CUDA(=HIP) tensors, enqueues hand-written gfx950 kernels on the current torch
stream through the C-ABI (``include/sad_amd.h``) and returns without synchronising.  There is no CPU
path: a CPU tensor raises ``RuntimeError``.  torch is used for device memory and streams only.
"""
import ctypes
import os
import sys
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from ._lib import MlpArgs, check, lib, vp


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _empty(*a, **k) -> torch.Tensor:
    """torch.empty; while a step plan is recorded (plan.py) the plan keeps the buffer, so its address stays valid for replays."""
    rec = _lib.recorder()
    return torch.empty(*a, **k) if rec is None else rec.empty(*a, **k)


def _unrecordable(what: str) -> None:
    """In front of any framework kernel on the step path (fill, strided copy): an error while a plan records, nothing otherwise."""
    if _lib.recorder() is not None:
        from .plan import PlanUnsupported
        raise PlanUnsupported(what)


def copy_rows(src: torch.Tensor, src_off_words: int, src_stride_words: int, n_rows: int, row_words: int,
              out: torch.Tensor, dst_stride_words: Optional[int] = None) -> torch.Tensor:
    """Strided row copy in 4-byte words through the library (``sad_copy_rows_u32``): row r of ``out`` = words
    [0, row_words) of row r of ``src`` starting at word ``src_off_words``, rows ``src_stride_words`` apart.  ``out`` must be
    contiguous.  The host side's replacement for framework copies on the step (recordable: plan.py)."""
    check(lib().sad_copy_rows_u32(src.data_ptr() + 4 * int(src_off_words), int(src_stride_words), out.data_ptr(),
                                  int(row_words if dst_stride_words is None else dst_stride_words), int(n_rows), int(row_words),
                                  _stream()), "sad_copy_rows_u32")
    return out


def copy_to_host(src: torch.Tensor, dst_pinned: torch.Tensor) -> None:
    """Device tensor -> PINNED host tensor of the same byte size, written by a kernel of the library over the bus (pinned
    memory is device-addressable; visible to the host once the launching stream has reached an event behind this call).
    For the few hundred KB a step hands back (boxes, NMS order and counts): a copy-engine transfer per tensor shared its
    queue with the step's H2D transfer and a few steps in a hundred stalled for milliseconds (tools/probe/pipeline_probe.py)."""
    if not src.is_cuda or not src.is_contiguous() or dst_pinned.is_cuda or not dst_pinned.is_pinned() or not dst_pinned.is_contiguous():
        raise TypeError("copy_to_host: expected a contiguous GPU source and a contiguous pinned host destination")
    nbytes = src.numel() * src.element_size()
    if nbytes != dst_pinned.numel() * dst_pinned.element_size() or nbytes % 4 != 0:
        raise ValueError("copy_to_host: sizes differ (or are not whole 4-byte words)")
    check(lib().sad_copy_rows_u32(src.data_ptr(), nbytes // 4, dst_pinned.data_ptr(), nbytes // 4, 1, nbytes // 4, _stream()),
          "sad_copy_rows_u32")


def split_points(points: torch.Tensor):
    """points [B,N,3+C] f32 contiguous -> (xyz [B,N,3], feat [B,N,C] or None), both packed (two library launches)."""
    B, N, D = points.shape
    xyz = _empty((B, N, 3), dtype=torch.float32, device=points.device)
    copy_rows(points, 0, D, B * N, 3, xyz)
    feat = None
    if D > 3:
        feat = _empty((B, N, D - 3), dtype=torch.float32, device=points.device)
        copy_rows(points, 3, D, B * N, D - 3, feat)
    return xyz, feat


def prefix_rows(x: torch.Tensor, m: int) -> torch.Tensor:
    """x [B,M,C] contiguous (C * element size a multiple of 4 bytes) -> packed copy of x[:, :m, :] (one library launch)."""
    B, M, C = x.shape
    rb = C * x.element_size()
    if rb % 4 != 0 or not x.is_contiguous() or not x.is_cuda:
        raise TypeError("prefix_rows: expected a contiguous GPU tensor whose rows are whole 4-byte words")
    out = _empty((B, m, C), dtype=x.dtype, device=x.device)
    return copy_rows(x, 0, M * rb // 4, B, m * rb // 4, out)


# Optional per-launch timing for bench.py: when LAUNCH_LOG is a list, every operator brackets its
# C-ABI call with HIP events recorded on the stream the kernel is launched on and appends
# (kind, name, start_event, end_event).  None (the default) adds nothing to the launch path.
LAUNCH_LOG: Optional[list] = None
# When a list, every MLP dispatch appends (name, fn): fn() enqueues the SAME dispatch again (same arguments; the
# tensors it reads are kept alive by the entry).  bench.py re-times the dispatches of a step back to back with it.
RERUN_LOG: Optional[list] = None

# ball_query_multi uses the grid-pruned kernel for scenes with at least this many points (scalar
# radii only); below it the brute-force scan is already cheap.  Set very large to force brute force.
GRID_MIN_POINTS: int = 2048

# When True, PackedMLP measures the workgroup geometries of mlp_chain_kernel on the first call for
# each shape and keeps the fastest (SADDetector.autotune() switches it on for one forward pass).
AUTOTUNE: bool = False


class _timed:
    __slots__ = ("kind", "name", "ev")

    def __init__(self, kind: str, name: str = ""):
        self.kind, self.name, self.ev = kind, name, None

    def __enter__(self):
        if LAUNCH_LOG is not None:
            self.ev = torch.cuda.Event(enable_timing=True)
            self.ev.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self.ev is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record(torch.cuda.current_stream())
            LAUNCH_LOG.append((self.kind, self.name, self.ev, end))
        return False


def _on_current_device(t: torch.Tensor, name: str) -> None:
    """The C-ABI launches on the CURRENT device's stream ("device chosen by caller"):
    a tensor that lives on another GPU would be dereferenced by a kernel running on the wrong one."""
    cur = torch.cuda.current_device()
    if t.device.index != cur:
        raise RuntimeError(f"{name}: tensor is on {t.device} but the current device is cuda:{cur}; "
                           f"wrap the call in `with torch.cuda.device({t.device.index}):`")


def _need(t: torch.Tensor, name: str, dtype, ndim: int) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (sad_amd has no CPU path)")
    _on_current_device(t, name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        _unrecordable(f"{name}: strided copy")
    return t.contiguous()


def fps(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    """Farthest point sampling (SPEC.md §2).  xyz [B,N,3] f32 -> idx [B,npoint] int32."""
    xyz = _need(xyz, "xyz", torch.float32, 3)
    B, N, three = xyz.shape
    if three != 3:
        raise ValueError("xyz: last dim must be 3")
    if not 1 <= npoint <= N:
        raise ValueError(f"npoint={npoint} must be in 1..N={N}")
    idx = _empty((B, npoint), dtype=torch.int32, device=xyz.device)
    ws_bytes = lib().sad_fps_workspace_bytes(B, N)
    ws = _empty((ws_bytes,), dtype=torch.uint8, device=xyz.device) if ws_bytes else None
    with _timed("fps", f"N{N}"):
        check(lib().sad_fps_f32(xyz.data_ptr(), B, N, npoint, idx.data_ptr(),
                                ws.data_ptr() if ws is not None else None, _stream()), "sad_fps_f32")
    return idx


def ffps(xyz: torch.Tensor, feat_pm: torch.Tensor, npoint: int, w_xyz: float = 1.0) -> torch.Tensor:
    """Feature-distance FPS (SPEC.md §15).  xyz [B,N,3] f32, feat_pm [B,N,C] f32 point-major ->
    idx [B,npoint] int32.  Allocates the B*N*N distance matrix the two-phase kernel needs."""
    xyz = _need(xyz, "xyz", torch.float32, 3)
    feat_pm = _need(feat_pm, "feat_pm", torch.float32, 3)
    B, N, _ = xyz.shape
    if feat_pm.shape[0] != B or feat_pm.shape[1] != N:
        raise ValueError("feat_pm must be [B,N,C]")
    if feat_pm.stride(2) != 1 or feat_pm.stride(0) != N * feat_pm.stride(1):
        feat_pm = feat_pm.contiguous()
    if not 1 <= npoint <= N:
        raise ValueError(f"npoint={npoint} must be in 1..N={N}")
    idx = _empty((B, npoint), dtype=torch.int32, device=xyz.device)
    ws = _empty((lib().sad_ffps_workspace_bytes(B, N),), dtype=torch.uint8, device=xyz.device)
    with _timed("fps", f"F{N}"):
        check(lib().sad_ffps_f32(xyz.data_ptr(), feat_pm.data_ptr(), feat_pm.stride(1), B, N,
                                 feat_pm.shape[2], npoint, float(w_xyz), idx.data_ptr(), ws.data_ptr(),
                                 _stream()), "sad_ffps_f32")
    return idx


def dfps_ffps(xyz: torch.Tensor, feat_pm: torch.Tensor, npoint: int, w_xyz: float = 1.0) -> torch.Tensor:
    """Fused D+F sampling (SPEC.md §15): npoint//2 picks by distance FPS, the rest by F-FPS."""
    half = npoint // 2
    parts = ([fps(xyz, half)] if half else []) + [ffps(xyz, feat_pm, npoint - half, w_xyz)]
    return torch.cat(parts, dim=1)


def gather_xyz(xyz: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """xyz [B,N,3], idx [B,M] -> [B,M,3] (SPEC.md §5)."""
    xyz = _need(xyz, "xyz", torch.float32, 3)
    idx = _need(idx, "idx", torch.int32, 2)
    B, N, _ = xyz.shape
    M = idx.shape[1]
    out = _empty((B, M, 3), dtype=torch.float32, device=xyz.device)
    check(lib().sad_gather_xyz_f32(xyz.data_ptr(), idx.data_ptr(), B, N, M, out.data_ptr(), _stream()),
          "sad_gather_xyz_f32")
    return out


def gather_points(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """features [B,C,N] (f32/bf16/f16), idx [B,M] -> [B,C,M] (SPEC.md §5)."""
    features = _need(features, "features", None, 3)
    idx = _need(idx, "idx", torch.int32, 2)
    esz = features.element_size()
    if esz not in (2, 4):
        raise TypeError("features: element size must be 2 or 4 bytes")
    B, C, N = features.shape
    M = idx.shape[1]
    out = _empty((B, C, M), dtype=features.dtype, device=features.device)
    check(lib().sad_gather_points(features.data_ptr(), idx.data_ptr(), B, C, N, M, esz,
                                  out.data_ptr(), _stream()), "sad_gather_points")
    return out


def group_points(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """features [B,C,N] (f32/bf16/f16), idx [B,M,S] -> [B,C,M,S] (SPEC.md §5)."""
    features = _need(features, "features", None, 3)
    idx = _need(idx, "idx", torch.int32, 3)
    esz = features.element_size()
    if esz not in (2, 4):
        raise TypeError("features: element size must be 2 or 4 bytes")
    B, C, N = features.shape
    _, M, S = idx.shape
    out = _empty((B, C, M, S), dtype=features.dtype, device=features.device)
    with _timed("group_points", f"C{C}N{N}M{M}S{S}"):
        check(lib().sad_group_points(features.data_ptr(), idx.data_ptr(), B, C, N, M, S, esz,
                                     out.data_ptr(), _stream()), "sad_group_points")
    return out


def subsample_pad(points: torch.Tensor, offsets: torch.Tensor, n_points: int, seed: int = 0,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Ragged scenes -> a fixed point count on the GPU (SPEC.md §17; the step before the path).
    points [total, C] f32 (all scenes concatenated), offsets [B+1] int32 on the GPU -> [B, n_points, C]:
    more points than needed = an evenly spread subset in file order, fewer = all points then hashed
    repeats, empty = zeros.  Same rows as the host loader ``io.fix_size(points_b, n_points, seed, scene=b)``."""
    points = _need(points, "points", torch.float32, 2)
    offsets = _need(offsets, "offsets", torch.int32, 1)
    B = offsets.shape[0] - 1
    if B < 1:
        raise ValueError("offsets must have B + 1 >= 2 entries")
    C = points.shape[1]
    if out is None:
        out = _empty((B, n_points, C), dtype=torch.float32, device=points.device)
    elif tuple(out.shape) != (B, n_points, C) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != points.device:
        raise ValueError(f"out: expected a contiguous float32 [{B},{n_points},{C}] tensor on {points.device}")
    check(lib().sad_subsample_pad_f32(points.data_ptr(), offsets.data_ptr(), B, C, int(n_points),
                                      int(seed) & 0xFFFFFFFF, out.data_ptr(), _stream()), "sad_subsample_pad_f32")
    return out


def ball_query(radius: Union[float, torch.Tensor], nsample: int, xyz: torch.Tensor,
               new_xyz: torch.Tensor) -> torch.Tensor:
    """Ball query (SPEC.md §3).  ``radius``: Python float (fixed) or [B,M] f32 tensor (adaptive,
    per centroid).  xyz [B,N,3], new_xyz [B,M,3] -> idx [B,M,nsample] int32."""
    xyz = _need(xyz, "xyz", torch.float32, 3)
    new_xyz = _need(new_xyz, "new_xyz", torch.float32, 3)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    if new_xyz.shape[0] != B:
        raise ValueError("xyz / new_xyz batch mismatch")
    idx = _empty((B, M, nsample), dtype=torch.int32, device=xyz.device)
    if isinstance(radius, torch.Tensor):
        rad = _need(radius, "radius", torch.float32, 2)
        if tuple(rad.shape) != (B, M):
            raise ValueError(f"radius tensor must be [B,M]=({B},{M})")
        check(lib().sad_ball_query_f32(xyz.data_ptr(), new_xyz.data_ptr(), 0.0, rad.data_ptr(), B, N, M,
                                       nsample, idx.data_ptr(), _stream()), "sad_ball_query_f32")
    else:
        check(lib().sad_ball_query_f32(xyz.data_ptr(), new_xyz.data_ptr(), float(np.float32(radius)),
                                       None, B, N, M, nsample, idx.data_ptr(), _stream()),
              "sad_ball_query_f32")
    return idx


def ball_query_multi(radii: Sequence[float], nsamples: Sequence[int], xyz: torch.Tensor,
                     new_xyz: torch.Tensor, radius_pc: Optional[torch.Tensor] = None,
                     return_counts: bool = False):
    """Several radii over the same (xyz, new_xyz): d2 is evaluated once per pair.  With
    ``radius_pc`` [B,M] the radius of branch r for centroid (b,m) is radii[r]*radius_pc[b,m]
    (SPEC.md §8 step 5).  Returns one idx [B,M,nsamples[r]] per radius; with ``return_counts`` also
    one int32 [B,M] per radius = accepted points capped at nsample (the non-padding rows of each
    group, which lets the fused MLP skip the padding without scanning idx)."""
    xyz = _need(xyz, "xyz", torch.float32, 3)
    new_xyz = _need(new_xyz, "new_xyz", torch.float32, 3)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    n = len(radii)
    if n != len(nsamples) or not 1 <= n <= _lib.MAX_RADII:
        raise ValueError(f"need 1..{_lib.MAX_RADII} radii with matching nsamples")
    outs = [_empty((B, M, s), dtype=torch.int32, device=xyz.device) for s in nsamples]
    r_arr = (ctypes.c_float * n)(*[float(np.float32(r)) for r in radii])
    s_arr = (ctypes.c_int * n)(*[int(s) for s in nsamples])
    p_arr = (vp * n)(*[o.data_ptr() for o in outs])
    cnts = [_empty((B, M), dtype=torch.int32, device=xyz.device) for _ in nsamples] if return_counts else None
    c_arr = (vp * n)(*[c.data_ptr() for c in cnts]) if return_counts else None
    pc = None
    if radius_pc is not None:
        radius_pc = _need(radius_pc, "radius_pc", torch.float32, 2)
        if tuple(radius_pc.shape) != (B, M):
            raise ValueError(f"radius_pc must be [B,M]=({B},{M})")
        pc = radius_pc.data_ptr()
    if pc is None and N >= GRID_MIN_POINTS and N <= 65536 and min(radii) > 0:
        # grid-pruned kernel: same indices, ~100x fewer pair tests (csrc/ball_query_grid.hip)
        ws = _empty((lib().sad_ball_query_grid_workspace_bytes(B, N),), dtype=torch.uint8,
                         device=xyz.device)
        with _timed("ball_query", f"N{N}M{M}x{n}"):
            check(lib().sad_ball_query_grid_f32(xyz.data_ptr(), new_xyz.data_ptr(), n, r_arr, s_arr, p_arr,
                                                c_arr, B, N, M, ws.data_ptr(), _stream()),
                  "sad_ball_query_grid_f32")
        return (outs, cnts) if return_counts else outs
    with _timed("ball_query", f"N{N}M{M}x{n}"):
        check(lib().sad_ball_query_multi_f32(xyz.data_ptr(), new_xyz.data_ptr(), n, r_arr, pc, s_arr,
                                             p_arr, c_arr, B, N, M, _stream()), "sad_ball_query_multi_f32")
    return (outs, cnts) if return_counts else outs


def rowscan_multi(idxs: Sequence[torch.Tensor], cnts: Sequence[torch.Tensor], N: int,
                  outs: Optional[Sequence[Tuple[torch.Tensor, int, int]]] = None) -> List[torch.Tensor]:
    """Row-packing tables (prefix sum of the per-group counts + row map) of up to four branches of one ball
    query, two launches for all of them.  Needs only what the ball query produced, so it can run on the stream
    that ran the query (the sampling stream), off the MLP stream's critical path; pass table i as the last
    element of branch i's ``grouped_multi`` call (``PackedMLP.grouped(..., ws=table)``).
    ``outs`` = [(out [B,M,ld_out] float32, col_off, C_out)] per branch: the scan also zero-fills the output slice of the
    groups the chain kernels combine with an atomic max, so ``out`` may be UNINITIALISED (``sad_mlp_rowscan_init``).
    Split pooling (bf16 mode): ``outs`` = [(out [B,M,ld_out] bfloat16, col_off, C_out, cont)] with ``cont`` from
    ``cont_buffer`` — nothing is filled, the tables carry what a split-pooled chain and the layer that reads its rows need
    (``sad_mlp_rowscan_split``)."""
    n = len(idxs)
    if n != len(cnts) or not 1 <= n <= _lib.MAX_RADII:
        raise ValueError(f"need 1..{_lib.MAX_RADII} (idx, cnt) pairs")
    B, M = cnts[0].shape
    wss = []
    for idx, cnt in zip(idxs, cnts):
        idx = _need(idx, "idx", torch.int32, 3)
        cnt = _need(cnt, "cnt", torch.int32, 2)
        if tuple(idx.shape[:2]) != (B, M) or tuple(cnt.shape) != (B, M):
            raise ValueError("idx / cnt shapes do not match")
        wss.append(_empty((lib().sad_mlp_workspace_bytes(B, M, idx.shape[2]),), dtype=torch.uint8, device=idx.device))
    c_arr = (vp * n)(*[c.data_ptr() for c in cnts])
    i_arr = (vp * n)(*[i.data_ptr() for i in idxs])
    s_arr = (ctypes.c_int * n)(*[int(i.shape[2]) for i in idxs])
    w_arr = (vp * n)(*[w.data_ptr() for w in wss])
    if outs is None:
        check(lib().sad_mlp_rowscan(n, c_arr, i_arr, s_arr, B, int(N), M, w_arr, _stream()), "sad_mlp_rowscan")
        return wss
    if len(outs) != n:
        raise ValueError("outs: one (out, col_off, C_out) per branch")
    if any(len(o) > 3 for o in outs):
        if not all(len(o) == 4 and o[3] is not None and o[0].dtype == torch.bfloat16 for o in outs):
            raise ValueError("outs: split pooling needs (bfloat16 out, col_off, C_out, cont) for EVERY branch of the scan")
        for (o, off, co, cont), idx in zip(outs, idxs):
            if tuple(o.shape[:2]) != (B, M) or not o.is_contiguous() or off < 0 or off + co > o.shape[2]:
                raise ValueError("outs: need contiguous [B,M,ld_out] buffers with col_off + C_out <= ld_out")
            if cont.numel() * cont.element_size() < lib().sad_mlp_cont_bytes(B, M, int(idx.shape[2]), int(co)):
                raise ValueError("outs: continuation buffer too small (ops.cont_buffer)")
        k_arr = (vp * n)(*[o[3].data_ptr() for o in outs])
        co_arr = (ctypes.c_int * n)(*[int(o[2]) for o in outs])
        check(lib().sad_mlp_rowscan_split(n, c_arr, i_arr, s_arr, B, int(N), M, w_arr, k_arr, co_arr, _stream()), "sad_mlp_rowscan_split")
        for w in wss:
            w._sad_split = True
        return wss
    for o, off, co in outs:
        o = _need(o, "out", torch.float32, 3)
        if tuple(o.shape[:2]) != (B, M) or not o.is_contiguous() or off < 0 or off + co > o.shape[2]:
            raise ValueError("outs: need contiguous [B,M,ld_out] float32 buffers with col_off + C_out <= ld_out")
    o_arr = (vp * n)(*[o.data_ptr() for o, _, _ in outs])
    ld_arr = (ctypes.c_int * n)(*[int(o.shape[2]) for o, _, _ in outs])
    off_arr = (ctypes.c_int * n)(*[int(off) for _, off, _ in outs])
    co_arr = (ctypes.c_int * n)(*[int(co) for _, _, co in outs])
    check(lib().sad_mlp_rowscan_init(n, c_arr, i_arr, s_arr, B, int(N), M, w_arr, o_arr, ld_arr, off_arr, co_arr, _stream()),
          "sad_mlp_rowscan_init")
    return wss


def cont_buffer(B: int, M: int, S: int, cout: int, device) -> torch.Tensor:
    """Continuation rows of one split-pooled bf16 chain (``sad_mlp_cont_bytes``; include/sad_amd.h ``sad_mlp_bf16_args.cont``)."""
    return _empty((lib().sad_mlp_cont_bytes(int(B), int(M), int(S), int(cout)),), dtype=torch.uint8, device=device)


_ITEMQ_INTS = 2 + 32 * 8      # csrc/common.h: a table carries the per-XCD item queues (and these ints) only when it has this many row starts


def workspace_status(ws: torch.Tensor, n_groups: Optional[int] = None) -> dict:
    """Instrumentation ints of a row-packing table (include/sad_amd.h, SAD_WS_*): weight-ring refills of the
    cooperative chain kernel, the id of the dispatch that owns the item queues right now and the conflict flag of the
    ``mlp_check_inuse`` knob.  Synchronises the device (a test / debugging helper, never on the measured path).
    ``n_groups`` = B * M of the table: a table with fewer than 258 row starts has no item queues (the kernels deal such
    launches statically) and ints 5..7 hold row starts there, so zeros are reported; pass it whenever the table may be small."""
    if n_groups is not None and n_groups + 1 < _ITEMQ_INTS:
        return {"refills": 0, "in_use": 0, "conflict": 0}
    torch.cuda.synchronize(ws.device)
    hdr = ws[:32].view(torch.int32).cpu()
    return {"refills": int(hdr[_lib.WS_REFILLS]), "in_use": int(hdr[_lib.WS_INUSE]), "conflict": int(hdr[_lib.WS_CONFLICT])}


def check_workspace(ws: torch.Tensor, n_groups: Optional[int] = None) -> None:
    """Raises if two dispatches were seen sharing ``ws`` at the same time (needs ``mlp_check_inuse=1``)."""
    st = workspace_status(ws, n_groups)
    if st["conflict"]:
        raise RuntimeError("row-packing workspace was used by two dispatches at the same time (one dispatch at a time per "
                           "workspace: sad_mlp_args.workspace in include/sad_amd.h)")


def knn_query(k: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """k nearest neighbours sorted by (d2, index) (SPEC.md §4).  -> idx [B,M,k] int32."""
    xyz = _need(xyz, "xyz", torch.float32, 3)
    new_xyz = _need(new_xyz, "new_xyz", torch.float32, 3)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = _empty((B, M, k), dtype=torch.int32, device=xyz.device)
    check(lib().sad_knn_f32(xyz.data_ptr(), new_xyz.data_ptr(), B, N, M, k, idx.data_ptr(), _stream()),
          "sad_knn_f32")
    return idx


def nms_bev_buffers(B: int, K: int, device) -> tuple:
    """(keep [B,K], order [B,K], count [B], workspace) for ``nms_bev(..., out=...)``: a caller that runs NMS every step
    allocates them once (pipeline.py)."""
    return (torch.empty((B, K), dtype=torch.int32, device=device), torch.empty((B, K), dtype=torch.int32, device=device),
            torch.empty((B,), dtype=torch.int32, device=device),
            torch.empty((lib().sad_nms_bev_workspace_bytes(B, K),), dtype=torch.uint8, device=device))


def nms_bev(boxes: torch.Tensor, iou_thr: float, score_thr: float = 0.0, single_kernel: bool = False, out: Optional[tuple] = None
            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Rotated-box NMS in bird's-eye view (SPEC.md §13).  boxes [B,K,9] f32, K <= 512 ->
    (keep [B,K] int32 0/1, order [B,K] int32 kept indices in rank order (-1 padded), count [B]).
    Default: the three-kernel variant (suppression matrix spread over the chip, workspace allocated
    here); ``single_kernel=True``: one workgroup per scene, no workspace.  Same keep decisions."""
    boxes = _need(boxes, "boxes", torch.float32, 3)
    B, K, nine = boxes.shape
    if nine != 9:
        raise ValueError("boxes: last dim must be 9 (x,y,z,l,w,h,yaw,score,label)")
    ws = None
    if out is not None:                 # buffers from nms_bev_buffers(B, K, device)
        keep, order, count, ws = out
        if tuple(keep.shape) != (B, K) or tuple(order.shape) != (B, K) or tuple(count.shape) != (B,) or keep.device != boxes.device:
            raise ValueError("out: buffers of another shape or device (nms_bev_buffers(B, K, device))")
    else:
        keep = _empty((B, K), dtype=torch.int32, device=boxes.device)
        order = _empty((B, K), dtype=torch.int32, device=boxes.device)
        count = _empty((B,), dtype=torch.int32, device=boxes.device)
    with _timed("nms", f"K{K}"):
        if single_kernel:
            check(lib().sad_nms_bev_f32(boxes.data_ptr(), B, K, float(np.float32(iou_thr)),
                                        float(np.float32(score_thr)), keep.data_ptr(), order.data_ptr(),
                                        count.data_ptr(), _stream()), "sad_nms_bev_f32")
        else:
            if ws is None:
                ws = _empty((lib().sad_nms_bev_workspace_bytes(B, K),), dtype=torch.uint8, device=boxes.device)
            check(lib().sad_nms_bev_ws_f32(boxes.data_ptr(), B, K, float(np.float32(iou_thr)),
                                           float(np.float32(score_thr)), keep.data_ptr(), order.data_ptr(),
                                           count.data_ptr(), ws.data_ptr(), _stream()), "sad_nms_bev_ws_f32")
    return keep, order, count


class PackedMLP:
    """A shared-MLP chain (SPEC.md §6) with weights repacked once into MFMA A-fragment order.

    ``layers`` = [(W [C_out,C_in], b [C_out]), ...] as numpy arrays or tensors (BatchNorm already
    folded).  ``first_has_xyz``: the first layer's input is [rel_xyz(3) ‖ features(C_in-3)].
    ``relu_mask`` bit l = ReLU after layer l (default: all layers).
    """

    def __init__(self, layers, first_has_xyz: bool, device, relu_mask: Optional[int] = None,
                 name: str = ""):
        self.name = name
        self._geom = {}              # shape key -> geometry picked by the autotuner
        # used for shapes never tuned (0 = built-in heuristic of the tiled kernel); see SADDetector.set_geometry
        self.default_geometry = 0
        # ... grouped calls that come with counts (cnt) and are not tuned use the kernel the library prefers for the shape
        # (register-resident / cooperative / layer-streamed chain; 0 = tiled): the un-tuned path is then within a few per
        # cent of the tuned one on the benchmark shapes
        self.preferred_geometry = 0
        if not 1 <= len(layers) <= _lib.MAX_LAYERS:
            raise ValueError(f"1..{_lib.MAX_LAYERS} layers supported")
        self.device = torch.device(device)
        ws, bs = [], []
        for w, b in layers:
            ws.append(torch.as_tensor(w, dtype=torch.float32).to(self.device).contiguous())
            bs.append(torch.as_tensor(b, dtype=torch.float32).to(self.device).contiguous())
        self.dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        for a, b in zip(ws[:-1], ws[1:]):
            if b.shape[1] != a.shape[0]:
                raise ValueError("layer shapes do not chain")
        self.L = len(ws)
        self.first_has_xyz = bool(first_has_xyz)
        self.relu_mask = (1 << self.L) - 1 if relu_mask is None else int(relu_mask)
        # A grouped 3-layer chain that is not a compiled shape of the register-resident kernels but is DOMINATED by one (every
        # width <= the shape's, at most 1.6 x the flops) is packed zero-padded onto that shape and runs there: `pack_dims` are the
        # dims the library sees, `dims` stay the chain's own (sad_mlp_args.c_out, ABI 3; DESIGN.md 6, generality table)
        self.pack_dims = list(self.dims)
        if self.first_has_xyz and self.L == 3 and self.relu_mask == (1 << self.L) - 1:
            pad_c = (ctypes.c_int * (self.L + 1))()
            if lib().sad_mlp_padded_dims(self.L, (ctypes.c_int * (self.L + 1))(*self.dims), pad_c):
                self.pack_dims = [int(v) for v in pad_c]
                for l in range(self.L):
                    w2 = torch.zeros((self.pack_dims[l + 1], self.pack_dims[l]), dtype=torch.float32, device=self.device)
                    w2[:ws[l].shape[0], :ws[l].shape[1]] = ws[l]
                    b2 = torch.zeros((self.pack_dims[l + 1],), dtype=torch.float32, device=self.device)
                    b2[:bs[l].shape[0]] = bs[l]
                    ws[l], bs[l] = w2.contiguous(), b2
        self.padded = self.pack_dims != self.dims
        dims_c = (ctypes.c_int * (self.L + 1))(*self.pack_dims)
        n = lib().sad_mlp_packed_floats(self.L, dims_c, int(self.first_has_xyz))
        self.packed = _empty((n,), dtype=torch.float32, device=self.device)
        w_arr = (vp * self.L)(*[w.data_ptr() for w in ws])
        b_arr = (vp * self.L)(*[b.data_ptr() for b in bs])
        with torch.cuda.device(self.device):
            check(lib().sad_mlp_pack_f32(self.L, dims_c, int(self.first_has_xyz), w_arr, b_arr,
                                         self.packed.data_ptr(), _stream()), "sad_mlp_pack_f32")
            torch.cuda.current_stream().synchronize()  # ws/bs may be freed after this returns
        self.out_channels = self.dims[-1]
        if self.first_has_xyz:
            self.preferred_geometry = int(lib().sad_mlp_preferred_geometry(self.L, dims_c))
        # geometry 3 (layer-streamed chain, csrc/mlp_layer.hip) applies when every layer's padded width is a
        # multiple of 128 channels; it needs scratch for the activations between layers
        # (plain rows: also C % 8 == 0 and an unpadded C_out, whole 128-channel blocks are stored)
        wide = all(((d + 31) // 32 * 32) % 128 == 0 for d in self.dims[1:])
        self._layered_ok = (not self.padded) and wide and (self.first_has_xyz or (self.dims[0] % 8 == 0 and self.dims[-1] % 128 == 0))

    # Geometries tried by the autotuner: W*100 + log2(WN)*10 + RW (include/sad_amd.h, sad_mlp_args).
    _CANDIDATES = [w * 100 + n * 10 + r for w in (8, 4) for n in range(4) if (1 << n) <= w
                   for r in (1, 2, 4)] + [1600 + n * 10 + r for n in (3, 4) for r in (1, 2)] \
        + [100000 + w * 100 + n * 10 + 1 for w in (8, 4) for n in range(3) if (2 << n) <= w] \
        + [200000 + w * 100 + n * 10 + 1 for w in (8, 4) for n in range(4) if (1 << n) <= w] \
        + [300000 + w * 100 + n * 10 + 1 for w in (8, 4) for n in range(3) if (2 << n) <= w]
    # +100000 flexible item distribution, +200000 two output tiles per wave, +300000 both
    _F_CODES = (2, 4, 5, 6)      # grouped mode: 2^f * R / S groups per workgroup (default f = 3)

    def _launch(self, a: MlpArgs, keep=None) -> None:
        """Enqueue the chain.  With AUTOTUNE on, the first call for a shape times every workgroup
        geometry that fits (a few ms, synchronous) and the fastest one is reused afterwards."""
        key = (bool(a.idx), a.B, a.N, a.M, a.S, a.ld_out)
        geom = self._geom.get(key)
        if geom is None and AUTOTUNE:
            geom = self._tune(a)
            self._geom[key] = geom
        a.geometry = geom or self.default_geometry or a.geometry      # (a.geometry: the preferred kernel of an un-tuned grouped call)
        with _timed("mlp", self.name):
            rc = lib().sad_mlp_chain_f32(ctypes.byref(a), _stream())
            if rc == -2 and not geom and not self.default_geometry and a.geometry > 5:
                # the library's un-tuned pick for the tiled kernel does not fit LDS for this (S, widths): its built-in heuristic
                self.preferred_geometry = 0
                a.geometry = 0
                rc = lib().sad_mlp_chain_f32(ctypes.byref(a), _stream())
            check(rc, "sad_mlp_chain_f32")
        if RERUN_LOG is not None:
            RERUN_LOG.append((self.name, lambda a=a, keep=keep: check(
                lib().sad_mlp_chain_f32(ctypes.byref(a), _stream()), "sad_mlp_chain_f32")))

    def _time(self, a: MlpArgs, stream) -> Optional[float]:
        """ms per launch of the geometry in ``a`` (None if it does not fit): one warm launch, then the
        faster of two timed batches of four — single batches of three picked different winners from
        run to run."""
        if lib().sad_mlp_chain_f32(ctypes.byref(a), _stream()) != 0:
            return None             # does not fit LDS / not valid for this nsample
        stream.synchronize()
        best = None
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(4):
                lib().sad_mlp_chain_f32(ctypes.byref(a), _stream())
            e1.record(stream)
            stream.synchronize()
            ms = e0.elapsed_time(e1) / 4
            best = ms if best is None or ms < best else best
        return best

    def _tune(self, a: MlpArgs) -> int:
        stream = torch.cuda.current_stream()
        best, best_ms = 0, None
        # 1 = VALU row-per-lane kernel (narrow chains), 2 = register-resident chain kernel (csrc/mlp_reg.hip),
        # 3 = layer-streamed chain (csrc/mlp_layer.hip), 4 = cooperative register-resident chain (csrc/mlp_coop.hip)
        # ... 5 = row-streaming plain layer (csrc/mlp_rows.hip)
        plain_extra = ([3] if self._layered_ok else []) + ([5] if self.L == 1 and self.dims[0] % 8 == 0 else [])
        for code in self._CANDIDATES + ([1] if a.idx else plain_extra):
            a.geometry = code
            ms = self._time(a, stream)
            if ms is not None and (best_ms is None or ms < best_ms * 0.98):   # prefer earlier entries on ties
                best, best_ms = code, ms
        # The kernels that consume a row-packing table (2 / 3 / 4) are timed apart and WIN unless the tiled kernel is more than
        # 10 % faster: a tiled pick for one chain of a stage costs what this timing does not see — its pooling buffer must be
        # zero-filled (a framework kernel: the step cannot be recorded into a plan any more, plan.py), its branch leaves the
        # stage's merged dispatch and its scan.  The cluster branch 259 -> 256 -> 256 -> 512 is within 2 - 3 % either way:
        # two of eleven runs of round 5 picked the tiled kernel for it and lost 7 % (f32), 8 % (the pipeline leg that shares
        # the geometry) of the step — very likely also the low readings round 4 could not explain.
        if a.idx and a.cnt and a.workspace:
            t_best, t_ms = 0, None
            for code in (2, 3, 4):
                a.geometry = code
                ms = self._time(a, stream)
                if ms is not None and (t_ms is None or ms < t_ms * 0.98):
                    t_best, t_ms = code, ms
            if os.environ.get("SAD_TUNE_DEBUG"):
                print(f"[tune] {self.name}: tiled {best} {best_ms}, table {t_best} {t_ms}", file=sys.stderr, flush=True)
            if t_ms is not None and (best_ms is None or t_ms <= best_ms * 1.10):
                return t_best
        if a.idx and best > 4:   # second sweep: groups per workgroup (how much padding is expected)
            base = best
            for f in self._F_CODES:
                a.geometry = base + 1000 * f
                ms = self._time(a, stream)
                if ms is not None and ms < best_ms * 0.98:
                    best, best_ms = base + 1000 * f, ms
            if a.cnt and a.workspace:   # third sweep: global row packing (1) vs per-workgroup packing (2)
                base, found = best, None
                for d in (1, 2):
                    a.geometry = base + 10000 * d
                    ms = self._time(a, stream)
                    if ms is not None and (found is None or ms < found[1]):
                        found = (base + 10000 * d, ms)
                if found is not None:
                    best = found[0]
        return best

    def _args(self) -> MlpArgs:
        a = MlpArgs()
        a.struct_size = ctypes.sizeof(MlpArgs)
        a.L = self.L
        for i, d in enumerate(self.pack_dims):
            a.dims[i] = d
        a.c_out = self.dims[-1] if self.padded else 0       # (a zero-padded chain stores only its own output channels)
        a.packed = self.packed.data_ptr()
        a.relu_mask = self.relu_mask
        return a

    @staticmethod
    def feat_fits_table_kernels(C: int, ld_feat: int, ptr: int) -> bool:
        """Can the register-resident / layer-streamed / cooperative kernels read feature rows of this layout?  They fetch
        16-byte chunks (C % 4 == 0, row stride % 4 == 0, 16-byte aligned base); a single strided channel (or none) is the
        other layout they take.  The ONE predicate behind ``wants_prescan`` and ``_grouped_args``: the first decides who
        prepares the pooling buffer, the second which kernel runs, and they must agree."""
        return C <= 1 or (C % 4 == 0 and ld_feat % 4 == 0 and ptr % 16 == 0)

    def wants_prescan(self, B: int, N: int, M: int, S: int, ld_out: int, C: int, feat: Optional[torch.Tensor] = None,
                      feat_dtype=torch.float32) -> bool:
        """Will a grouped call of this shape (with counts) run a kernel that consumes a caller-made row-packing table
        (geometries 2 / 3 / 4)?  The tiled kernel packs with its own tile height: a table made for it would be wasted —
        and it expects a ZERO pooling buffer, which the scan does not give it.  ``feat``: the feature tensor the call will
        get ([B,N,C] point-major, any stride); None = a fresh contiguous float32 [B,N,C] tensor (a stage output)."""
        geom = self._geom.get((True, B, N, M, S, ld_out)) or self.default_geometry
        if feat is None:
            fits = self.feat_fits_table_kernels(C, C, 0)
        else:
            fits = (feat.dim() == 3 and feat.stride(2) == 1 and feat.stride(0) == N * feat.stride(1)
                    and self.feat_fits_table_kernels(feat.shape[2], feat.stride(1), feat.data_ptr()))
        if not geom and not AUTOTUNE and fits:
            geom = self.preferred_geometry
        return geom % 1000 in (2, 3, 4)

    def grouped(self, xyz: torch.Tensor, feat_pm: Optional[torch.Tensor], new_xyz: torch.Tensor,
                idx: torch.Tensor, out: Optional[torch.Tensor] = None, col_off: int = 0,
                cnt: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Fused group -> MLP -> max over nsample.  xyz [B,N,3]; feat_pm point-major [B,N,C] (or
        None); new_xyz [B,M,3]; idx [B,M,S].  Writes out[:, :, col_off:col_off+C_out] of a
        point-major [B,M,ld_out] buffer (allocated [B,M,C_out] when ``out`` is None).  A caller-
        provided ``out`` slice must be ZERO on entry (groups spanning two row tiles are combined
        with an atomic max).  Samples that repeat a group's first index are skipped.  ``ws``: the
        row-packing table of (idx, cnt) from ``rowscan_multi`` (else the chain scans the counts itself)."""
        a, out, _keep = self._grouped_args(xyz, feat_pm, new_xyz, idx, out, col_off, cnt, ws)
        self._launch(a, _keep + [out])
        return out

    def _grouped_args(self, xyz, feat_pm, new_xyz, idx, out, col_off, cnt, ws=None):
        """Validated ``MlpArgs`` of a grouped call + the output tensor + tensors to keep alive."""
        if not self.first_has_xyz:
            raise RuntimeError("this PackedMLP was packed without the xyz prefix")
        xyz = _need(xyz, "xyz", torch.float32, 3)
        new_xyz = _need(new_xyz, "new_xyz", torch.float32, 3)
        idx = _need(idx, "idx", torch.int32, 3)
        B, N, _ = xyz.shape
        _, M, S = idx.shape
        a = self._args()
        keep = [xyz, new_xyz, idx]
        feat_ok16 = True
        if feat_pm is None:
            C = 0
        else:
            if not feat_pm.is_cuda or feat_pm.dtype != torch.float32 or feat_pm.dim() != 3:
                raise TypeError("feat_pm: expected a GPU float32 [B,N,C] tensor")
            if feat_pm.stride(2) != 1 or feat_pm.stride(0) != N * feat_pm.stride(1):
                _unrecordable("feat_pm: strided copy")
                feat_pm = feat_pm.contiguous()
            C = feat_pm.shape[2]
            a.feat = feat_pm.data_ptr()
            a.ld_feat = feat_pm.stride(1)
            keep.append(feat_pm)
            # (the register-resident / layer-streamed kernels read feature rows as 16-byte chunks; a single strided
            # channel is the other layout they take)
            feat_ok16 = self.feat_fits_table_kernels(C, a.ld_feat, feat_pm.data_ptr())
        if self.dims[0] != C + 3:
            raise ValueError(f"MLP expects {self.dims[0] - 3} feature channels, got {C}")
        if self.padded:
            # a zero-padded chain runs on the register-resident kernels only: they take counts and 16-byte feature rows
            if cnt is None:
                _unrecordable("padded chain: counts derived from idx")
                pos = torch.arange(1, S + 1, device=idx.device, dtype=torch.int32)
                cnt = torch.clamp(((idx != idx[..., :1]).to(torch.int32) * pos).amax(-1), min=1).to(torch.int32).contiguous()
            if feat_pm is not None and not feat_ok16:
                _unrecordable("padded chain: packed copy of the features")
                feat_pm = feat_pm.contiguous()
                a.feat, a.ld_feat = feat_pm.data_ptr(), feat_pm.stride(1)
                keep.append(feat_pm)
                feat_ok16 = self.feat_fits_table_kernels(C, a.ld_feat, feat_pm.data_ptr())
        if out is None:   # the kernel max-combines into the buffer: it must start at zero
            _unrecordable("grouped: zero-filled output")
            out = torch.zeros((B, M, self.out_channels), dtype=torch.float32, device=xyz.device)
        if self.relu_mask != (1 << self.L) - 1:
            raise RuntimeError("grouped chains need a ReLU after every layer (max-pool combine)")
        self._check_out(out, B * M, col_off)
        a.xyz, a.new_xyz, a.idx = xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr()
        if cnt is not None:   # [B,M] int32 from ball_query_multi(return_counts=True)
            cnt = _need(cnt, "cnt", torch.int32, 2)
            if tuple(cnt.shape) != (B, M):
                raise ValueError("cnt must be [B,M]")
            a.cnt = cnt.data_ptr()
            if ws is None:
                ws = _empty((lib().sad_mlp_workspace_bytes(B, M, S),), dtype=torch.uint8, device=xyz.device)
            else:             # table already filled by rowscan_multi (geometries 2 / 3 then launch no scan)
                if ws.numel() < lib().sad_mlp_workspace_bytes(B, M, S):
                    raise ValueError("ws: too small for this (B, M, S)")
                a.prescanned = 1
            a.workspace = ws.data_ptr()   # global row packing
            keep += [cnt, ws]
        a.B, a.N, a.M, a.S, a.C = B, N, M, S, C
        a.out, a.ld_out, a.col_off = out.data_ptr(), out.stride(-2), col_off
        a.geometry = self._geom.get((bool(a.idx), a.B, a.N, a.M, a.S, a.ld_out)) or self.default_geometry
        if not a.geometry and cnt is not None and not AUTOTUNE and feat_ok16:
            a.geometry = self.preferred_geometry
        if a.prescanned and a.geometry % 1000 not in (2, 3, 4):
            # backstop: a caller-made table means the caller's scan prepared `out` for a table kernel (only the groups those
            # kernels combine atomically were zeroed); the tiled kernel packs for itself and max-combines into memory it
            # expects to be zero — never run it on such a buffer
            raise RuntimeError(f"{self.name or 'PackedMLP'}: a row-packing table (ws) was passed but geometry {a.geometry} packs for "
                               "itself; ask wants_prescan(..., feat=<the feature tensor>) before making the table")
        if self._layered_ok and cnt is not None and (a.geometry == 3 or AUTOTUNE):
            dims_c = (ctypes.c_int * (self.L + 1))(*self.pack_dims)
            nbytes = lib().sad_mlp_scratch_bytes(B, M, S, self.L, dims_c)
            sc = _empty((nbytes,), dtype=torch.uint8, device=xyz.device)
            a.scratch, a.scratch_bytes = sc.data_ptr(), nbytes
            keep.append(sc)
        return a, out, keep

    def rows(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, col_off: int = 0
             ) -> torch.Tensor:
        """Plain rows.  x [..., C] point-major (last-dim stride 1) -> [..., C_out]."""
        if self.first_has_xyz:
            raise RuntimeError("this PackedMLP was packed with the xyz prefix")
        if not x.is_cuda or x.dtype != torch.float32:
            raise TypeError("x: expected a GPU float32 tensor")
        _on_current_device(x, "x")
        C = x.shape[-1]
        if C != self.dims[0]:
            raise ValueError(f"MLP expects {self.dims[0]} channels, got {C}")
        if not x.is_contiguous():
            _unrecordable("rows: strided input")
        x2 = x.reshape(-1, C)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        R = x2.shape[0]
        if out is None:
            out = _empty(tuple(x.shape[:-1]) + (self.out_channels,), dtype=torch.float32,
                              device=x.device)
        self._check_out(out, R, col_off)
        a = self._args()
        a.feat, a.ld_feat = x2.data_ptr(), x2.stride(0)
        a.B, a.N, a.M, a.S, a.C = 1, 0, R, 1, C
        a.out, a.ld_out, a.col_off = out.data_ptr(), out.stride(-2), col_off
        geom = self._geom.get((bool(a.idx), a.B, a.N, a.M, a.S, a.ld_out)) or self.default_geometry
        sc = None
        if self._layered_ok and self.L > 1 and (geom == 3 or AUTOTUNE):   # activations between the layer launches
            dims_c = (ctypes.c_int * (self.L + 1))(*self.dims)
            nbytes = lib().sad_mlp_scratch_bytes(1, R, 1, self.L, dims_c)
            sc = _empty((nbytes,), dtype=torch.uint8, device=x.device)
            a.scratch, a.scratch_bytes = sc.data_ptr(), nbytes
        self._launch(a, [x2, out, sc])
        return out

    def _check_out(self, out: torch.Tensor, rows: int, col_off: int) -> None:
        if not out.is_cuda or out.dtype != torch.float32 or out.stride(-1) != 1:
            raise TypeError("out: expected a GPU float32 tensor with unit last-dim stride")
        if out.numel() // out.shape[-1] != rows or not out.is_contiguous():
            raise ValueError("out: expected a contiguous [rows, ld_out] buffer")
        if col_off < 0 or col_off + self.out_channels > out.shape[-1]:
            raise ValueError("out: col_off + C_out exceeds the buffer width")


# bf16 chains: merge the branches of a stage into one dispatch (sad_mlp_chain_multi_bf16)?
MERGE_BF16: bool = True
# bf16 stages with an aggregation layer: split pooling (bf16 pooled rows + continuation rows, no atomics, no zero fill; sa_module.can_split)?
SPLIT_POOL: bool = not os.environ.get("SAD_NO_SPLIT_POOL")


def grouped_multi(calls) -> None:
    """Several independent fused group -> MLP -> max launches (the branches of one multi-radius
    stage) as ONE dispatch (``sad_mlp_chain_multi_f32``): the light chains fill the tail of the
    heavy one.  ``calls`` = [(PackedMLP, xyz, feat_pm, new_xyz, idx, out, col_off, cnt[, ws]), ...] with
    caller-provided zero ``out`` buffers (``ws``: row-packing table from ``rowscan_multi``).  While autotuning, or for a single call, each chain is
    launched (and tuned) on its own."""
    if AUTOTUNE or len(calls) < 2 or (not MERGE_BF16 and isinstance(calls[0][0], PackedMLPBf16)):
        for c in calls:
            mlp, xyz, feat_pm, new_xyz, idx, out, col_off, cnt = c[:8]
            if len(c) > 9 and c[9] is not None:
                mlp.grouped(xyz, feat_pm, new_xyz, idx, out=out, col_off=col_off, cnt=cnt, ws=c[8], cont=c[9])
            else:
                mlp.grouped(xyz, feat_pm, new_xyz, idx, out=out, col_off=col_off, cnt=cnt)
        if AUTOTUNE and len(calls) >= 2:
            _tune_stage([c[:8] for c in calls])
        return
    args, keep = [], []
    for c in calls:
        mlp, xyz, feat_pm, new_xyz, idx, out, col_off, cnt = c[:8]
        ws = c[8] if len(c) > 8 else None
        if len(c) > 9 and c[9] is not None:      # (split pooling: bf16 chains only)
            a, _, k = mlp._grouped_args(xyz, feat_pm, new_xyz, idx, out, col_off, cnt, ws, cont=c[9])
        else:
            a, _, k = mlp._grouped_args(xyz, feat_pm, new_xyz, idx, out, col_off, cnt, ws)
        args.append(a)
        keep.append(k)
    bf16 = isinstance(calls[0][0], PackedMLPBf16)
    if any(isinstance(c[0], PackedMLPBf16) != bf16 for c in calls):
        raise TypeError("grouped_multi: all chains must be of the same class (f32 or bf16)")
    _rec = _lib.recorder()
    if _rec is not None:             # (a recorded step replays this dispatch: its argument blocks live as long as the plan)
        _rec.keep.append((args, keep))
    with _timed("mlp", "+".join(c[0].name for c in calls)):
        if bf16:
            arr = (ctypes.POINTER(_lib.MlpBf16Args) * len(args))(*[ctypes.pointer(a) for a in args])
            check(lib().sad_mlp_chain_multi_bf16(arr, len(args), _stream()), "sad_mlp_chain_multi_bf16")
            if RERUN_LOG is not None:
                outs = [c[5] for c in calls]
                RERUN_LOG.append(("+".join(c[0].name for c in calls), lambda arr=arr, args=args, keep=(keep, outs): check(
                    lib().sad_mlp_chain_multi_bf16(arr, len(args), _stream()), "sad_mlp_chain_multi_bf16")))
        else:
            arr = (ctypes.POINTER(MlpArgs) * len(args))(*[ctypes.pointer(a) for a in args])
            rc = lib().sad_mlp_chain_multi_f32(arr, len(args), _stream())
            if rc == -2 and any(a.geometry > 5 and not c[0]._geom and not c[0].default_geometry for a, c in zip(args, calls)):
                for a, c in zip(args, calls):      # an un-tuned tiled pick that does not fit LDS here: the built-in heuristic
                    if a.geometry > 5 and not c[0]._geom and not c[0].default_geometry:
                        c[0].preferred_geometry = 0
                        a.geometry = 0
                rc = lib().sad_mlp_chain_multi_f32(arr, len(args), _stream())
            check(rc, "sad_mlp_chain_multi_f32")
            if RERUN_LOG is not None:
                outs = [c[5] for c in calls]
                RERUN_LOG.append(("+".join(c[0].name for c in calls), lambda arr=arr, args=args, keep=(keep, outs): check(
                    lib().sad_mlp_chain_multi_f32(arr, len(args), _stream()), "sad_mlp_chain_multi_f32")))


def choose_stage_assignment(picked, t_picked, uniform_times, table=(2, 3, 4)):
    """The stage-level decision of the autotuner as a pure function (CPU-testable).  ``picked`` = per-chain codes with the measured
    time ``t_picked`` of their merged dispatch; ``uniform_times`` = {table code: time of the dispatch with every chain on it, or
    None when refused}.  Among the uniform assignments the fastest wins (an earlier code keeps a tie within 2 %).  It replaces
    per-chain picks that are all table kernels when it is 2 % faster — and picks with a TILED kernel among them unless those are
    more than 10 % faster: a tiled branch needs its pooling slice zero-filled by a framework kernel (the step can no longer be
    replayed from a plan), leaves the stage's scan and its merged dispatch, none of which this timing sees."""
    u_best, u_t = None, None
    for code in table:
        t = uniform_times.get(code)
        if t is not None and (u_t is None or t < u_t * 0.98):
            u_best, u_t = code, t
    all_table = all(p in table for p in picked)
    if u_t is not None:
        if (all_table and (t_picked is None or u_t < t_picked * 0.98)) or (not all_table and (t_picked is None or u_t <= t_picked * 1.10)):
            return [u_best] * len(picked), u_t
    return list(picked), t_picked


def _tune_stage(calls) -> None:
    """Stage-level autotune step: the branches of a stage go out as ONE dispatch, and register-resident
    (geometry 2) / layer-streamed (geometry 3) chains share their launches and work lists, so a branch
    that is slower on its own (few tiles) may still be best inside the merged dispatch.  Times the merged
    dispatch with the per-chain picks against all-2 and all-3 and keeps the fastest assignment."""
    stream = torch.cuda.current_stream()
    bf16 = isinstance(calls[0][0], PackedMLPBf16)
    fn = lib().sad_mlp_chain_multi_bf16 if bf16 else lib().sad_mlp_chain_multi_f32
    args_cls = _lib.MlpBf16Args if bf16 else MlpArgs
    keys = []
    for mlp, xyz, feat_pm, new_xyz, idx, out, col_off, cnt in calls:
        a, _, _ = mlp._grouped_args(xyz, feat_pm, new_xyz, idx, out, col_off, cnt)
        keys.append((bool(a.idx), a.B, a.N, a.M, a.S, a.ld_out))

    def run(codes):
        args, keep = [], []
        for (mlp, xyz, feat_pm, new_xyz, idx, out, col_off, cnt), code in zip(calls, codes):
            a, _, k = mlp._grouped_args(xyz, feat_pm, new_xyz, idx, out, col_off, cnt)
            a.geometry = code
            args.append(a)
            keep.append(k)
        arr = (ctypes.POINTER(args_cls) * len(args))(*[ctypes.pointer(a) for a in args])
        if fn(arr, len(args), _stream()) != 0:
            return None
        stream.synchronize()
        best = None
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(4):
                fn(arr, len(args), _stream())
            e1.record(stream)
            stream.synchronize()
            ms = e0.elapsed_time(e1) / 4
            best = ms if best is None or ms < best else best
        return best

    picked = [c[0]._geom.get(k) or 0 for c, k in zip(calls, keys)]
    table = (2,) if bf16 else (2, 3, 4)          # the kernels whose chains share launches (f32: register-resident, layer-streamed, cooperative)
    all_table = all(p in table for p in picked)
    t_picked = run(picked)
    uniform = {}
    for code in table:
        uniform[code] = t_picked if (all_table and all(p == code for p in picked)) else run([code] * len(calls))
    best, t_best = choose_stage_assignment(picked, t_picked, uniform, table)
    if os.environ.get("SAD_TUNE_DEBUG"):
        print(f"[tune-stage] {'+'.join(c[0].name for c in calls)}: picked {picked} {t_picked}, uniform {uniform}, final {best} {t_best}", file=sys.stderr, flush=True)
    for c, k, code in zip(calls, keys, best):
        c[0]._geom[k] = code


class PackedMLPBf16:
    """The same chain in bfloat16 on the matrix cores (SPEC.md §14, BASELINE.json configs[4]).

    Weights are rounded to bf16 once and stored in MFMA fragment order; features are bf16 tensors
    (float32 accepted and rounded on load); accumulation is float32.  Grouped output is float32
    (pooled), plain output float32 or bfloat16.  Dense rows — ball-query padding is computed."""

    def __init__(self, layers, first_has_xyz: bool, device, relu_mask: Optional[int] = None,
                 name: str = ""):
        self.name = name
        if not 1 <= len(layers) <= _lib.MAX_LAYERS:
            raise ValueError(f"1..{_lib.MAX_LAYERS} layers supported")
        self.device = torch.device(device)
        ws = [torch.as_tensor(w, dtype=torch.float32).to(self.device).contiguous() for w, _ in layers]
        bs = [torch.as_tensor(b, dtype=torch.float32).to(self.device).contiguous() for _, b in layers]
        self.dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        for a, b in zip(ws[:-1], ws[1:]):
            if b.shape[1] != a.shape[0]:
                raise ValueError("layer shapes do not chain")
        self.L = len(ws)
        self.first_has_xyz = bool(first_has_xyz)
        self.relu_mask = (1 << self.L) - 1 if relu_mask is None else int(relu_mask)
        dims_c = (ctypes.c_int * (self.L + 1))(*self.dims)
        n = lib().sad_mlp_packed_bytes_bf16(self.L, dims_c, int(self.first_has_xyz))
        self.packed = _empty((n,), dtype=torch.uint8, device=self.device)
        self._geom = {}      # (mode, B, N, M, S, ld_out) -> rows per tile picked by the autotuner
        self.default_geometry = 0
        w_arr = (vp * self.L)(*[w.data_ptr() for w in ws])
        b_arr = (vp * self.L)(*[b.data_ptr() for b in bs])
        with torch.cuda.device(self.device):
            check(lib().sad_mlp_pack_bf16(self.L, dims_c, int(self.first_has_xyz), w_arr, b_arr,
                                          self.packed.data_ptr(), _stream()), "sad_mlp_pack_bf16")
            torch.cuda.current_stream().synchronize()
        self.out_channels = self.dims[-1]
        # grouped calls that come with counts use the register-resident chain kernel (geometry 2, csrc/mlp_bf16_reg.hip)
        # where the library has the shape compiled
        self.preferred_geometry = int(lib().sad_mlp_preferred_geometry_bf16(self.L, dims_c)) if self.first_has_xyz else 0

    def _feat_ok_reg(self, feat_pm) -> bool:
        if feat_pm is None:
            return True
        C = feat_pm.shape[2]
        if C <= 13:
            return True
        return (feat_pm.dtype == torch.bfloat16 and C % 8 == 0 and feat_pm.stride(1) % 8 == 0 and feat_pm.data_ptr() % 16 == 0)

    def wants_prescan(self, B: int, N: int, M: int, S: int, ld_out: int, C: int, feat: Optional[torch.Tensor] = None,
                      feat_dtype=torch.bfloat16) -> bool:
        """See ``PackedMLP.wants_prescan``: geometry 2 consumes a caller-made row-packing table.  ``feat`` None = a fresh
        contiguous [B,N,C] tensor of ``feat_dtype`` (a stage output); the same ``_feat_ok_reg`` rule decides in ``_grouped_args``."""
        geom = self._geom.get((True, B, N, M, S, ld_out)) or self.default_geometry
        if feat is None:
            fits = C <= 13 or (feat_dtype == torch.bfloat16 and C % 8 == 0)
        else:
            fits = (feat.dim() == 3 and feat.stride(2) == 1 and feat.stride(0) == N * feat.stride(1) and self._feat_ok_reg(feat))
        if not geom and not AUTOTUNE and fits:
            geom = self.preferred_geometry
        return geom == 2

    def _args(self) -> _lib.MlpBf16Args:
        a = _lib.MlpBf16Args()
        a.struct_size = ctypes.sizeof(_lib.MlpBf16Args)
        a.L = self.L
        for i, d in enumerate(self.dims):
            a.dims[i] = d
        a.packed = self.packed.data_ptr()
        a.relu_mask = self.relu_mask
        return a

    @staticmethod
    def _feat(t: torch.Tensor, name: str):
        if not t.is_cuda or t.dtype not in (torch.bfloat16, torch.float32):
            raise TypeError(f"{name}: expected a GPU bfloat16 or float32 tensor")
        return 1 if t.dtype == torch.bfloat16 else 0

    def grouped(self, xyz: torch.Tensor, feat_pm: Optional[torch.Tensor], new_xyz: torch.Tensor,
                idx: torch.Tensor, out: Optional[torch.Tensor] = None, col_off: int = 0,
                cnt: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
                cont: Optional[torch.Tensor] = None) -> torch.Tensor:
        """xyz [B,N,3] f32; feat_pm point-major [B,N,C] bf16/f32 (or None); new_xyz [B,M,3]; idx
        [B,M,S] -> out[:, :, col_off:col_off+C_out] of a ZERO-initialised float32 [B,M,ld] buffer.
        With ``cnt`` ([B,M] int32 from ball_query_multi(return_counts=True)) only the leading cnt
        rows of each group are computed — the ball query's padding rows cannot change the max.
        ``ws``: the row-packing table of (idx, cnt) from ``rowscan_multi`` (geometry 2 then launches no scan).
        Split pooling: ``out`` an UNINITIALISED bfloat16 [B,M,ld] buffer and ``cont`` from ``cont_buffer`` (needs ``cnt`` and the
        register-resident chain) — the true pooled row is the maximum of out[g] and the group's continuation rows, which
        ``rows(..., pool=...)`` takes while it reads them (include/sad_amd.h ``sad_mlp_bf16_args.cont``)."""
        a, out, _keep = self._grouped_args(xyz, feat_pm, new_xyz, idx, out, col_off, cnt, ws, cont)
        self._launch(a, _keep + [out])
        return out

    def _grouped_args(self, xyz, feat_pm, new_xyz, idx, out, col_off, cnt, ws=None, cont=None):
        """Validated ``MlpBf16Args`` of a grouped call + the output tensor + tensors to keep alive."""
        if not self.first_has_xyz:
            raise RuntimeError("this PackedMLPBf16 was packed without the xyz prefix")
        xyz = _need(xyz, "xyz", torch.float32, 3)
        new_xyz = _need(new_xyz, "new_xyz", torch.float32, 3)
        idx = _need(idx, "idx", torch.int32, 3)
        B, N, _ = xyz.shape
        _, M, S = idx.shape
        a = self._args()
        keep = [xyz, new_xyz, idx]
        C = 0
        if feat_pm is not None:
            a.feat_bf16 = self._feat(feat_pm, "feat_pm")
            if feat_pm.dim() != 3:
                raise ValueError("feat_pm: expected [B,N,C]")
            if feat_pm.stride(2) != 1 or feat_pm.stride(0) != N * feat_pm.stride(1):
                _unrecordable("feat_pm: strided copy")
                feat_pm = feat_pm.contiguous()
            C = feat_pm.shape[2]
            a.feat, a.ld_feat = feat_pm.data_ptr(), feat_pm.stride(1)
            keep.append(feat_pm)
        if self.dims[0] != C + 3:
            raise ValueError(f"MLP expects {self.dims[0] - 3} feature channels, got {C}")
        if out is None:
            _unrecordable("grouped: zero-filled output")
            out = torch.zeros((B, M, self.out_channels), dtype=torch.float32, device=xyz.device)
        split = out.dtype == torch.bfloat16
        if (not split and out.dtype != torch.float32) or not out.is_contiguous() or col_off + self.out_channels > out.shape[-1]:
            raise ValueError("out: expected a contiguous float32 (or, split pooling, bfloat16) [B,M,ld_out] buffer wide enough")
        if split != (cont is not None):
            raise ValueError("split pooling needs both a bfloat16 out and a continuation buffer (ops.cont_buffer)")
        a.xyz, a.new_xyz, a.idx = xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr()
        a.B, a.N, a.M, a.S, a.C = B, N, M, S, C
        a.out, a.out_bf16, a.ld_out, a.col_off = out.data_ptr(), int(split), out.stride(-2), col_off
        if split:
            if cnt is None or self.preferred_geometry != 2 or not self._feat_ok_reg(feat_pm):
                raise RuntimeError(f"{self.name or 'PackedMLPBf16'}: split pooling runs on the register-resident chain only (cnt, a compiled shape, 16-byte bf16 feature rows)")
            if cont.numel() * cont.element_size() < lib().sad_mlp_cont_bytes(B, M, S, self.out_channels):
                raise ValueError("cont: too small (ops.cont_buffer)")
            if ws is not None and not getattr(ws, "_sad_split", False):
                raise RuntimeError("split pooling: the row-packing table must come from rowscan_multi with split-pooling outs")
            a.cont = cont.data_ptr()
            keep.append(cont)
        if cnt is not None:
            cnt = _need(cnt, "cnt", torch.int32, 2)
            if tuple(cnt.shape) != (B, M):
                raise ValueError("cnt must be [B,M]")
            given = ws is not None
            if not given:
                ws = _empty((lib().sad_mlp_workspace_bytes(B, M, S),), dtype=torch.uint8, device=xyz.device)
            elif ws.numel() < lib().sad_mlp_workspace_bytes(B, M, S):
                raise ValueError("ws: too small for this (B, M, S)")
            a.cnt, a.workspace = cnt.data_ptr(), ws.data_ptr()
            keep += [cnt, ws]
        a.geometry = self._geom.get((bool(a.idx), a.B, a.N, a.M, a.S, a.ld_out)) or self.default_geometry
        if not a.geometry and cnt is not None and not AUTOTUNE and self._feat_ok_reg(feat_pm):
            a.geometry = self.preferred_geometry
        if split:
            a.geometry = 2
        if cnt is not None and given:
            if a.geometry != 2:       # (same backstop as PackedMLP._grouped_args: the tiled kernel needs a ZERO buffer)
                raise RuntimeError(f"{self.name or 'PackedMLPBf16'}: a row-packing table (ws) was passed but geometry {a.geometry} "
                                   "packs for itself; ask wants_prescan(..., feat=<the feature tensor>) before making the table")
            a.prescanned = 1
        return a, out, keep

    def _launch(self, a, keep=None) -> None:
        """Enqueue; with AUTOTUNE on, the first call for a shape times 64 / 128 / 256 rows per tile.  ``keep``: tensors the
        launch reads or writes (kept alive by a RERUN_LOG entry)."""
        key = (bool(a.idx), a.B, a.N, a.M, a.S, a.ld_out)
        geom = self._geom.get(key)
        preferred = a.geometry          # the un-tuned choice of _grouped_args (0 while autotuning)
        if a.idx and a.out_bf16:        # split pooling: the register-resident chain, nothing to tune
            with _timed("mlp", self.name):
                check(lib().sad_mlp_chain_bf16(ctypes.byref(a), _stream()), "sad_mlp_chain_bf16")
            if RERUN_LOG is not None:
                RERUN_LOG.append((self.name, lambda a=a, keep=keep: check(
                    lib().sad_mlp_chain_bf16(ctypes.byref(a), _stream()), "sad_mlp_chain_bf16")))
            return
        if geom is None and AUTOTUNE:
            stream = torch.cuda.current_stream()
            best, best_ms, reg_ms = 0, None, None
            for code in (0, 32, 64, 128, 256) + ((2,) if a.cnt and a.workspace and not a.prescanned else ()):
                a.geometry = code
                if lib().sad_mlp_chain_bf16(ctypes.byref(a), _stream()) != 0:
                    continue
                stream.synchronize()
                ms_best = None
                for _ in range(2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(4):
                        lib().sad_mlp_chain_bf16(ctypes.byref(a), _stream())
                    e1.record(stream)
                    stream.synchronize()
                    ms = e0.elapsed_time(e1)
                    ms_best = ms if ms_best is None or ms < ms_best else ms_best
                if code == 2:
                    reg_ms = ms_best
                elif best_ms is None or ms_best < best_ms * 0.98:
                    best, best_ms = code, ms_best
            # (the register-resident chain wins unless the tiled kernel is more than 10 % faster: see PackedMLP._tune)
            if reg_ms is not None and (best_ms is None or reg_ms <= best_ms * 1.10):
                best = 2
            geom = best
            self._geom[key] = geom
            preferred = 0
        a.geometry = geom or self.default_geometry or preferred
        with _timed("mlp", self.name):
            check(lib().sad_mlp_chain_bf16(ctypes.byref(a), _stream()), "sad_mlp_chain_bf16")
        if RERUN_LOG is not None:
            RERUN_LOG.append((self.name, lambda a=a, keep=keep: check(
                lib().sad_mlp_chain_bf16(ctypes.byref(a), _stream()), "sad_mlp_chain_bf16")))

    def takes_pooled(self, rows: int, ld_out: int) -> bool:
        """Can ``rows(..., pool=...)`` read split-pooled rows: one layer on the row-streaming kernel (the autotuner may have picked a
        tiled kernel for this layer: then not)."""
        geom = self._geom.get((False, 1, 0, rows, 1, ld_out)) or self.default_geometry
        return self.L == 1 and not self.first_has_xyz and self.dims[0] % 8 == 0 and geom in (0, 3)

    def rows(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, col_off: int = 0,
             out_dtype=torch.float32, pool=None) -> torch.Tensor:
        """Plain rows.  x [..., C] bf16/f32 (last-dim stride 1) -> [..., C_out] f32 or bf16.
        ``pool`` = [(ws, cont, S, cols)] per chain, in column order: ``x`` holds SPLIT-POOLED rows (``grouped(..., cont=...)``)
        of these chains side by side; the layer takes the maximum with their continuation rows while it reads them."""
        if self.first_has_xyz:
            raise RuntimeError("this PackedMLPBf16 was packed with the xyz prefix")
        a = self._args()
        a.feat_bf16 = self._feat(x, "x")
        keep_pool = []
        if pool:
            if len(pool) > _lib.MAX_RADII or x.dtype != torch.bfloat16 or not x.is_contiguous():
                raise ValueError(f"pool: at most {_lib.MAX_RADII} chains behind contiguous bfloat16 rows")
            a.n_pool = len(pool)
            for i, (ws, cont, S, cols) in enumerate(pool):
                a.pool_ws[i], a.pool_cont[i], a.pool_S[i], a.pool_cols[i] = ws.data_ptr(), cont.data_ptr(), int(S), int(cols)
                keep_pool += [ws, cont]
        C = x.shape[-1]
        if C != self.dims[0]:
            raise ValueError(f"MLP expects {self.dims[0]} channels, got {C}")
        if not x.is_contiguous():
            _unrecordable("rows: strided input")
        x2 = x.reshape(-1, C)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        R = x2.shape[0]
        if out is None:
            out = _empty(tuple(x.shape[:-1]) + (self.out_channels,), dtype=out_dtype, device=x.device)
        if out.dtype not in (torch.float32, torch.bfloat16) or not out.is_contiguous() \
                or out.numel() // out.shape[-1] != R or col_off + self.out_channels > out.shape[-1]:
            raise ValueError("out: expected a contiguous f32/bf16 [rows, ld_out] buffer wide enough")
        a.feat, a.ld_feat = x2.data_ptr(), x2.stride(0)
        a.B, a.N, a.M, a.S, a.C = 1, 0, R, 1, C
        a.out, a.out_bf16 = out.data_ptr(), int(out.dtype == torch.bfloat16)
        a.ld_out, a.col_off = out.stride(-2), col_off
        if pool:
            a.geometry = 0                # (the row-streaming layer: the only reader of split-pooled rows)
            with _timed("mlp", self.name):
                check(lib().sad_mlp_chain_bf16(ctypes.byref(a), _stream()), "sad_mlp_chain_bf16")
            if RERUN_LOG is not None:
                RERUN_LOG.append((self.name, lambda a=a, keep=[x2, out] + keep_pool: check(
                    lib().sad_mlp_chain_bf16(ctypes.byref(a), _stream()), "sad_mlp_chain_bf16")))
            _rec = _lib.recorder()
            if _rec is not None:
                _rec.keep.append((a, keep_pool))
            return out
        self._launch(a, [x2, out])
        return out


def mlp_chain(x: torch.Tensor, layers, relu_mask: Optional[int] = None) -> torch.Tensor:
    """One-shot convenience: pack ``layers`` and apply them to rows x [..., C]."""
    return PackedMLP(layers, False, x.device, relu_mask).rows(x)
