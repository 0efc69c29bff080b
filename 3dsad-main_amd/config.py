"""Model topology and work accounting for the SA + size-adaptive-clustering path.

The upstream reference ships no config (``/root/reference/README.md:1-2`` is the whole repository);
the topology is this repository's SPEC.md §11, frozen from SURVEY.md §8(d).
Pure Python / numpy only — importable without torch and without a GPU.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

ANCHOR_CAR = (3.9, 1.6, 1.56)
ANCHORS = ((3.9, 1.6, 1.56), (0.8, 0.6, 1.73), (1.76, 0.6, 1.73))
SHIFT_MAX = 2.0
R_MIN = 1.0
R_MAX = 4.8


@dataclass(frozen=True)
class SAStage:
    """One multi-radius set-abstraction stage (SPEC.md §7)."""
    npoint: int
    radii: Tuple[float, ...]
    nsamples: Tuple[int, ...]
    mlps: Tuple[Tuple[int, ...], ...]   # per branch, output channels of each layer
    agg: int                            # aggregation 1x1 conv output channels (0 = none)


@dataclass(frozen=True)
class DetectorConfig:
    """SPEC.md §11."""
    n_points: int = 16384
    in_feat: int = 1                    # intensity
    stages: Tuple[SAStage, ...] = (
        SAStage(4096, (0.2, 0.4, 0.8), (32, 32, 64), ((16, 16, 32), (16, 16, 32), (32, 32, 64)), 64),
        SAStage(1024, (0.4, 0.8, 1.6), (32, 32, 64), ((64, 64, 128), (64, 64, 128), (64, 96, 128)), 128),
        SAStage(512, (1.6, 3.2, 4.8), (32, 32, 32), ((128, 128, 256), (128, 192, 256), (128, 256, 256)), 256),
    )
    n_cand: int = 256
    cand_mlp: Tuple[int, ...] = (128, 6)
    cluster_scales: Tuple[float, ...] = (1.0, 2.0)
    cluster_nsamples: Tuple[int, ...] = (16, 32)
    cluster_mlps: Tuple[Tuple[int, ...], ...] = ((256, 256, 512), (256, 512, 1024))
    cluster_agg: int = 512
    head_mlp: Tuple[int, ...] = (256, 256, 10)
    anchor_car: Tuple[float, ...] = ANCHOR_CAR      # SPEC.md §8
    anchors: Tuple[Tuple[float, ...], ...] = ANCHORS  # SPEC.md §9
    shift_max: float = SHIFT_MAX
    r_min: float = R_MIN
    r_max: float = R_MAX


KITTI = DetectorConfig()

# A small topology with the same structure, used by CPU tests / smoke so the oracle finishes in
# seconds.  Same code paths (3 MSG stages, adaptive cluster layer, head), smaller sizes.
TINY = DetectorConfig(
    n_points=2048,
    stages=(
        SAStage(512, (0.8, 1.6, 3.2), (32, 32, 64), ((16, 16, 32), (16, 16, 32), (32, 32, 64)), 64),
        SAStage(256, (1.6, 3.2, 6.4), (32, 32, 64), ((64, 64, 128), (64, 64, 128), (64, 96, 128)), 128),
        SAStage(128, (3.2, 6.4, 9.6), (32, 32, 32), ((128, 128, 256), (128, 192, 256), (128, 256, 256)), 256),
    ),
    n_cand=64,
)

# BASELINE.json configs[4]: nuScenes-shaped scenes — 65 536 points on [-51.2, 51.2]^2 with 4 extra
# channels, "same topology x4 points" (SURVEY.md §8(d)); run with dtype="bf16" (SPEC.md §14).
NUSCENES = DetectorConfig(
    n_points=65536,
    in_feat=4,
    stages=(
        SAStage(16384, (0.2, 0.4, 0.8), (32, 32, 64), ((16, 16, 32), (16, 16, 32), (32, 32, 64)), 64),
        SAStage(4096, (0.4, 0.8, 1.6), (32, 32, 64), ((64, 64, 128), (64, 64, 128), (64, 96, 128)), 128),
        SAStage(2048, (1.6, 3.2, 4.8), (32, 32, 32), ((128, 128, 256), (128, 192, 256), (128, 256, 256)), 256),
    ),
    n_cand=1024,
)

# BASELINE.json configs[0]: one SA layer, no features, no aggregation.
CONFIG0_SA = SAStage(256, (0.2,), (32,), ((64, 64, 128),), 0)


def mlp_layers(cfg: DetectorConfig) -> List[Tuple[str, List[int]]]:
    """Every MLP chain of the detector as (name, [C_in, C_1, ..., C_L]) in forward order."""
    out = []
    c = cfg.in_feat
    for si, st in enumerate(cfg.stages):
        for bi, mlp in enumerate(st.mlps):
            out.append((f"sa{si + 1}.b{bi}", [c + 3] + list(mlp)))
        cat = sum(m[-1] for m in st.mlps)
        if st.agg:
            out.append((f"sa{si + 1}.agg", [cat, st.agg]))
            c = st.agg
        else:
            c = cat
    out.append(("cand", [c] + list(cfg.cand_mlp)))
    for bi, mlp in enumerate(cfg.cluster_mlps):
        out.append((f"cluster.b{bi}", [c + 3] + list(mlp)))
    cat = sum(m[-1] for m in cfg.cluster_mlps)
    out.append(("cluster.agg", [cat, cfg.cluster_agg]))
    out.append(("head", [cfg.cluster_agg] + list(cfg.head_mlp)))
    return out


def work_per_scene(cfg: DetectorConfig) -> dict:
    """Algorithmic work of one scene (what roofline fractions are computed from)."""
    n = cfg.n_points
    c = cfg.in_feat
    fps_updates = 0
    fps_steps = 0
    pair_tests = 0
    bq_bytes = 0
    group_bytes = 0
    flops = 0
    rows = {}
    for si, st in enumerate(cfg.stages):
        m = st.npoint
        fps_updates += n * m
        fps_steps += m
        for r, s, mlp in zip(st.radii, st.nsamples, st.mlps):
            pair_tests += n * m
            bq_bytes += n * 12 + m * 12 + m * s * 4
            group_bytes += m * s * 4 + (c + 3) * n * 4 + (c + 3) * m * s * 4
            dims = [c + 3] + list(mlp)
            flops += 2 * m * s * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        cat = sum(mm[-1] for mm in st.mlps)
        if st.agg:
            flops += 2 * m * cat * st.agg
            c = st.agg
        else:
            c = cat
        n = m
    k = cfg.n_cand
    dims = [c] + list(cfg.cand_mlp)
    flops += 2 * k * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    for s, mlp in zip(cfg.cluster_nsamples, cfg.cluster_mlps):
        pair_tests += n * k
        bq_bytes += n * 12 + k * 12 + k * 4 + k * s * 4
        group_bytes += k * s * 4 + (c + 3) * n * 4 + (c + 3) * k * s * 4
        dims = [c + 3] + list(mlp)
        flops += 2 * k * s * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    cat = sum(mm[-1] for mm in cfg.cluster_mlps)
    flops += 2 * k * cat * cfg.cluster_agg
    dims = [cfg.cluster_agg] + list(cfg.head_mlp)
    flops += 2 * k * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    return dict(fps_updates=fps_updates, fps_steps=fps_steps, pair_tests=pair_tests,
                ball_query_bytes=bq_bytes, group_points_bytes=group_bytes, mlp_flops=flops)
