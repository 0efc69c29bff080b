"""Seeded synthetic scenes and weights (SPEC.md §12).  numpy only.

The reference ships no data, loaders or checkpoints (``/root/reference/README.md:1-2``); there is
no network either, so every measurement in this repository runs on these synthetic inputs.
"""
import numpy as np

from .config import DetectorConfig, mlp_layers


def make_scene(scene_id: int, n_points: int = 16384, extent=(0.0, 70.4, -40.0, 40.0),
               n_boxes: int = 40) -> np.ndarray:
    """One KITTI-shaped scene -> float32 [n_points, 4] = (x, y, z, intensity)."""
    rng = np.random.default_rng(1234 + scene_id)
    x0, x1, y0, y1 = extent
    n_obj = int(round(0.3 * n_points))
    n_gnd = n_points - n_obj
    gx = rng.uniform(x0, x1, n_gnd)
    gy = rng.uniform(y0, y1, n_gnd)
    gz = rng.normal(-1.6, 0.1, n_gnd)
    # car-sized boxes standing on the ground plane
    bc = np.stack([rng.uniform(x0 + 3, x1 - 3, n_boxes), rng.uniform(y0 + 3, y1 - 3, n_boxes)], 1)
    yaw = rng.uniform(-np.pi, np.pi, n_boxes)
    which = rng.integers(0, n_boxes, n_obj)
    loc = rng.uniform(-0.5, 0.5, (n_obj, 3)) * np.array([3.9, 1.6, 1.56])
    c, s = np.cos(yaw[which]), np.sin(yaw[which])
    ox = bc[which, 0] + c * loc[:, 0] - s * loc[:, 1]
    oy = bc[which, 1] + s * loc[:, 0] + c * loc[:, 1]
    oz = -1.6 + 0.78 + loc[:, 2]
    pts = np.stack([np.concatenate([gx, ox]), np.concatenate([gy, oy]), np.concatenate([gz, oz]),
                    rng.uniform(0.0, 1.0, n_points)], 1)
    pts = pts[rng.permutation(n_points)]
    return np.ascontiguousarray(pts.astype(np.float32))


def make_batch(first_scene: int, batch: int, n_points: int = 16384, **kw) -> np.ndarray:
    """float32 [batch, n_points, 4]; scene ids first_scene .. first_scene+batch-1."""
    return np.stack([make_scene(first_scene + i, n_points, **kw) for i in range(batch)], 0)


def make_nuscenes_batch(first_scene: int, batch: int, n_points: int = 65536) -> np.ndarray:
    """BASELINE.json configs[4] input: float32 [batch, n_points, 7] — the KITTI recipe on
    [-51.2, 51.2]^2 with 160 boxes, plus 3 more uniform channels (4 extra channels in all)."""
    out = []
    for i in range(batch):
        base = make_scene(first_scene + i, n_points, extent=(-51.2, 51.2, -51.2, 51.2), n_boxes=160)
        extra = np.random.default_rng(99991 + first_scene + i).uniform(0.0, 1.0, (n_points, 3)).astype(np.float32)
        out.append(np.concatenate([base, extra], 1))
    return np.ascontiguousarray(np.stack(out, 0))


def make_tiny_batch(first_scene: int, batch: int, n_points: int = 2048) -> np.ndarray:
    """Small dense scenes for the TINY topology (20 m x 20 m, 6 boxes)."""
    return make_batch(first_scene, batch, n_points, extent=(0.0, 20.0, -10.0, 10.0), n_boxes=6)


def make_dense_batch(first_scene: int, batch: int, n_points: int = 16384) -> np.ndarray:
    """Dense-occupancy variant of the KITTI-shaped workload (bench.py ``--scene dense``): the same
    16 384 points on the TINY extents (20 m x 20 m, 6 boxes), ~40 points per square metre instead of
    ~3, so the wider ball queries find nsample neighbours and few grouped rows are padding."""
    return make_batch(first_scene, batch, n_points, extent=(0.0, 20.0, -10.0, 10.0), n_boxes=6)


def make_unit_cube(scene_id: int, n_points: int = 1024) -> np.ndarray:
    """BASELINE.json configs[0] input: float32 [n_points, 3] uniform in the unit cube."""
    rng = np.random.default_rng(1234 + scene_id)
    return np.ascontiguousarray(rng.uniform(0.0, 1.0, (n_points, 3)).astype(np.float32))


def make_mlp_weights(dims, rng) -> list:
    """[(W[C_out,C_in], b[C_out]), ...] Kaiming-uniform weights, small uniform biases."""
    out = []
    for cin, cout in zip(dims[:-1], dims[1:]):
        bound = np.sqrt(6.0 / cin)
        w = rng.uniform(-bound, bound, (cout, cin)).astype(np.float32)
        b = rng.uniform(-0.1, 0.1, (cout,)).astype(np.float32)
        out.append((np.ascontiguousarray(w), np.ascontiguousarray(b)))
    return out


def make_weights(cfg: DetectorConfig, seed: int = 0) -> dict:
    """name -> [(W, b), ...] for every MLP chain of the detector (config.mlp_layers order)."""
    rng = np.random.default_rng(seed)
    w = {name: make_mlp_weights(dims, rng) for name, dims in mlp_layers(cfg)}
    # The two regression outputs feed clamps (SPEC.md §8-§9).  Plain Kaiming init saturates every
    # clamp (all radii = r_max), which would leave the adaptive-radius path untested, so the last
    # layer of those two chains is scaled down to keep shifts / sizes inside their ranges.
    for name, scale in (("cand", 0.05), ("head", 0.1)):
        W, b = w[name][-1]
        w[name][-1] = (np.ascontiguousarray((W * np.float32(scale)).astype(np.float32)), b)
    return w
