"""Point-cloud file formats on the input side of the path: KITTI velodyne
``.bin`` (N x 4 float32: x, y, z, intensity) and nuScenes ``.pcd.bin`` (N x 5 float32: x, y, z,
intensity, ring), plus the deterministic subsample / pad that turns a ragged scene into the fixed
``n_points`` rows the kernels take (a batch has one N: SPEC.md conventions).
The upstream reference (``/root/reference/README.md:1-2``) ships no loader; formats are the public
dataset layouts.  numpy only (memory-mapped reads: files are streamed, not loaded whole).
"""
import os
from typing import Iterable, Optional, Tuple

import numpy as np

KITTI_COLS = 4
NUSCENES_COLS = 5


def read_bin(path: str, cols: int = KITTI_COLS) -> np.ndarray:
    """Memory-map a little-endian float32 point file -> read-only [N, cols] view."""
    size = os.path.getsize(path)
    if size % (4 * cols) != 0:
        raise ValueError(f"{path}: {size} bytes is not a whole number of {cols}-float points")
    if size == 0:                       # (an empty file cannot be memory-mapped)
        return np.zeros((0, cols), dtype="<f4")
    return np.memmap(path, dtype="<f4", mode="r", shape=(size // (4 * cols), cols))


def crop_range(points: np.ndarray, extent: Tuple[float, float, float, float, float, float]
               ) -> np.ndarray:
    """Keep points with x0 <= x < x1, y0 <= y < y1, z0 <= z < z1 (KITTI: 0,70.4,-40,40,-3,1)."""
    x0, x1, y0, y1, z0, z1 = extent
    p = np.asarray(points)
    m = ((p[:, 0] >= x0) & (p[:, 0] < x1) & (p[:, 1] >= y0) & (p[:, 1] < y1) &
         (p[:, 2] >= z0) & (p[:, 2] < z1))
    return p[m]


def _mix(a: np.ndarray) -> np.ndarray:
    """SPEC.md §17 ``mix`` on uint32 arrays (wrapping arithmetic)."""
    a = a.astype(np.uint32)
    a ^= a >> np.uint32(16)
    a = (a.astype(np.uint64) * np.uint64(0x85EBCA6B)).astype(np.uint32)
    a ^= a >> np.uint32(13)
    a = (a.astype(np.uint64) * np.uint64(0xC2B2AE35)).astype(np.uint32)
    a ^= a >> np.uint32(16)
    return a


def _h(seed: int, scene: int, i) -> np.ndarray:
    """SPEC.md §17 ``h(b, i)`` for an array (or scalar) of row numbers ``i``."""
    i = np.asarray(i, dtype=np.uint64)
    a = (np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x9E3779B1) + np.uint64(scene & 0xFFFFFFFF) * np.uint64(0x85EBCA77)
         + i * np.uint64(0xC2B2AE3D) + np.uint64(0x27D4EB2F)) & np.uint64(0xFFFFFFFF)
    return _mix(a.astype(np.uint32))


def select_rows(n: int, n_points: int, seed: int = 0, scene: int = 0) -> np.ndarray:
    """SPEC.md §17: source row (0..n-1) of every output row of a scene with ``n`` > 0 points."""
    i = np.arange(n_points, dtype=np.int64)
    if n == n_points:
        return i
    if n > n_points:      # systematic subsample with a hashed start: strictly increasing rows
        r = int(_h(seed, scene, 0xFFFFFFFF)) % n
        return (i * n + r) // n_points
    pad = (_h(seed, scene, i).astype(np.int64)) % n
    return np.where(i < n, i, pad)


def fix_size(points: np.ndarray, n_points: int, seed: int = 0, scene: int = 0) -> np.ndarray:
    """Ragged scene -> exactly ``n_points`` rows, deterministically for (seed, scene): SPEC.md §17, the
    rule ``ops.subsample_pad`` (HIP) and the oracle implement bit for bit.  More points than needed:
    an evenly spread subset in file order (systematic sampling, hashed start).  Fewer: all points,
    then hashed repeats of existing points (duplicates are harmless to fps / ball_query / max-pool).
    An empty scene becomes all zeros."""
    p = np.ascontiguousarray(points, dtype=np.float32)
    n = p.shape[0]
    if n == n_points:
        return p
    if n == 0:
        return np.zeros((n_points, p.shape[1]), np.float32)
    return np.ascontiguousarray(p[select_rows(n, n_points, seed, scene)])


def load_scene(path: str, n_points: int, cols: int = KITTI_COLS, use_cols: int = 4,
               extent: Optional[Tuple[float, ...]] = None, seed: int = 0, scene: int = 0) -> np.ndarray:
    """One file -> float32 [n_points, use_cols] (x, y, z, intensity by default)."""
    pts = read_bin(path, cols)[:, :use_cols]
    if extent is not None:
        pts = crop_range(pts, extent)
    return fix_size(pts, n_points, seed, scene)


def load_batch(paths: Iterable[str], n_points: int, **kw) -> np.ndarray:
    """Files -> float32 [B, n_points, use_cols]; file i is scene i of SPEC.md §17."""
    seed = kw.pop("seed", 0)
    return np.stack([load_scene(p, n_points, seed=seed, scene=i, **kw) for i, p in enumerate(paths)], 0)


def load_ragged(paths: Iterable[str], cols: int = KITTI_COLS, use_cols: int = 4,
                extent: Optional[Tuple[float, ...]] = None):
    """Files -> (points [sum N_i, use_cols] float32, offsets [B+1] int32): the ragged input of
    ``ops.subsample_pad`` (the subsample / pad then runs on the GPU)."""
    parts, offs = [], [0]
    for p in paths:
        pts = read_bin(p, cols)[:, :use_cols]
        if extent is not None:
            pts = crop_range(pts, extent)
        parts.append(np.ascontiguousarray(pts, dtype=np.float32))
        offs.append(offs[-1] + parts[-1].shape[0])
    return (np.concatenate(parts, 0) if parts else np.zeros((0, use_cols), np.float32)), np.asarray(offs, np.int32)
