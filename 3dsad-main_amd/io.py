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
    return np.memmap(path, dtype="<f4", mode="r", shape=(size // (4 * cols), cols))


def crop_range(points: np.ndarray, extent: Tuple[float, float, float, float, float, float]
               ) -> np.ndarray:
    """Keep points with x0 <= x < x1, y0 <= y < y1, z0 <= z < z1 (KITTI: 0,70.4,-40,40,-3,1)."""
    x0, x1, y0, y1, z0, z1 = extent
    p = np.asarray(points)
    m = ((p[:, 0] >= x0) & (p[:, 0] < x1) & (p[:, 1] >= y0) & (p[:, 1] < y1) &
         (p[:, 2] >= z0) & (p[:, 2] < z1))
    return p[m]


def fix_size(points: np.ndarray, n_points: int, seed: int = 0) -> np.ndarray:
    """Ragged scene -> exactly ``n_points`` rows, deterministically for a given seed.
    More points than needed: a uniform random subset, kept in file order.  Fewer: all points, then
    random repeats of existing points (duplicates are harmless to fps / ball_query / max-pool).
    An empty scene becomes all zeros."""
    p = np.ascontiguousarray(points, dtype=np.float32)
    n = p.shape[0]
    if n == n_points:
        return p
    rng = np.random.default_rng(seed)
    if n > n_points:
        sel = np.sort(rng.choice(n, n_points, replace=False))
        return np.ascontiguousarray(p[sel])
    if n == 0:
        return np.zeros((n_points, p.shape[1]), np.float32)
    extra = rng.integers(0, n, n_points - n)
    return np.ascontiguousarray(np.concatenate([p, p[extra]], 0))


def load_scene(path: str, n_points: int, cols: int = KITTI_COLS, use_cols: int = 4,
               extent: Optional[Tuple[float, ...]] = None, seed: int = 0) -> np.ndarray:
    """One file -> float32 [n_points, use_cols] (x, y, z, intensity by default)."""
    pts = read_bin(path, cols)[:, :use_cols]
    if extent is not None:
        pts = crop_range(pts, extent)
    return fix_size(pts, n_points, seed)


def load_batch(paths: Iterable[str], n_points: int, **kw) -> np.ndarray:
    """Files -> float32 [B, n_points, use_cols]; scene i is subsampled with seed + i."""
    seed = kw.pop("seed", 0)
    return np.stack([load_scene(p, n_points, seed=seed + i, **kw) for i, p in enumerate(paths)], 0)
