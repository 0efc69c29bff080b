"""Regenerates the golden vectors under tests/golden/ from the CPU spec-oracle.

Run from the repo root:  python tests/golden/make_golden.py

PARITY UNPINNED: the upstream reference (/root/reference/README.md:1-2) ships no implementation,
tests or fixtures, so these vectors come from this repository's own oracle (oracle/sad_oracle.c,
restating SPEC.md).  They pin the oracle against accidental change and are what the GPU box checks
the HIP kernels against without needing to trust a freshly built oracle.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import sad_amd  # noqa: E402,F401
from sad_amd import config, synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def config0():
    """BASELINE.json configs[0]: 1 scene x 1024 pts, npoint=256, r=0.2, nsample=32, mlp [64,64,128]."""
    xyz = synth.make_unit_cube(0, 1024)[None]
    st = config.CONFIG0_SA
    rng = np.random.default_rng(7)
    layers = synth.make_mlp_weights([3, 64, 64, 128], rng)
    fidx = oracle.fps(xyz, st.npoint)
    new_xyz = oracle.gather_xyz(xyz, fidx)
    bidx = oracle.ball_query(st.radii[0], st.nsamples[0], xyz, new_xyz)
    feat = oracle.sa_group_mlp_max(xyz, None, new_xyz, bidx, layers)
    knn = oracle.knn_query(16, xyz, new_xyz)
    np.savez_compressed(os.path.join(OUT, "config0.npz"), xyz=xyz, fps_idx=fidx, new_xyz=new_xyz,
                        ball_idx=bidx, knn_idx=knn, feat=feat.astype(np.float32))


def edge_cases():
    """Hand-checkable toys: collinear, duplicates, empty ball, > nsample in ball, exact ties."""
    d = {}
    # collinear points on the x axis at 0,1,2,...,9 ; exact ties in distance
    line = np.zeros((1, 10, 3), np.float32)
    line[0, :, 0] = np.arange(10)
    d["line_xyz"] = line
    d["line_fps5"] = oracle.fps(line, 5)
    cen = np.array([[[4.5, 0, 0], [0, 0, 0], [100, 0, 0]]], np.float32)
    d["line_cen"] = cen
    d["line_bq_r2_s4"] = oracle.ball_query(2.0, 4, line, cen)      # >nsample, ties, empty ball
    d["line_bq_r1.5_s8"] = oracle.ball_query(1.5, 8, line, cen)    # padding with first index
    d["line_knn3"] = oracle.knn_query(3, line, cen)                # 4 and 5 tie for centre 4.5
    # duplicates: 6 copies of one point + 2 distinct
    dup = np.zeros((1, 8, 3), np.float32)
    dup[0, 6] = (1, 0, 0)
    dup[0, 7] = (0, 2, 0)
    d["dup_xyz"] = dup
    d["dup_fps6"] = oracle.fps(dup, 6)
    d["dup_bq"] = oracle.ball_query(0.5, 4, dup, dup[:, :3].copy())
    # strict compare: a point exactly at distance r is outside
    sq = np.array([[[0, 0, 0], [3, 4, 0], [0, 5, 0], [0.5, 0, 0]]], np.float32)
    d["strict_xyz"] = sq
    d["strict_bq"] = oracle.ball_query(5.0, 4, sq, sq[:, :1].copy())
    np.savez_compressed(os.path.join(OUT, "edge_cases.npz"), **d)


def adaptive():
    """Per-centroid adaptive radius on a KITTI-shaped scene."""
    pts = synth.make_tiny_batch(100, 2, 2048)
    xyz = np.ascontiguousarray(pts[:, :, :3])
    fidx = oracle.fps(xyz, 128)
    new_xyz = oracle.gather_xyz(xyz, fidx)
    rng = np.random.default_rng(11)
    rad = rng.uniform(0.3, 3.0, (2, 128)).astype(np.float32)
    idx = oracle.ball_query(rad, 16, xyz, new_xyz)
    np.savez_compressed(os.path.join(OUT, "adaptive.npz"), fps_idx=fidx, radius=rad, ball_idx=idx)


def tiny_detector():
    """TINY topology end to end: 2 scenes -> boxes, plus the index decisions on the way."""
    cfg = config.TINY
    pts = synth.make_tiny_batch(0, 2, cfg.n_points)
    w = synth.make_weights(cfg, 0)
    tr = {}
    boxes = oracle.detector_forward(pts, cfg, w, tr)
    np.savez_compressed(
        os.path.join(OUT, "tiny_detector.npz"), boxes=boxes,
        sa1_fps=tr["sa1"]["fps_idx"], sa2_fps=tr["sa2"]["fps_idx"], sa3_fps=tr["sa3"]["fps_idx"],
        sa3_out=tr["sa3"]["out"], radius=tr["cluster"]["radius"], cand=tr["cluster"]["cand"],
        cl_idx0=tr["cluster"]["ball_idx"][0], cl_idx1=tr["cluster"]["ball_idx"][1],
        head=tr["cluster"]["head"])


def extensions():
    """SPEC.md §13-§16 operators: NMS, bf16 chain, F-FPS, scatter-add / arg-max references."""
    rng = np.random.default_rng(1316)
    d = {}
    xyz = rng.random((2, 400, 3), dtype=np.float32)
    feat = rng.standard_normal((2, 400, 6)).astype(np.float32)
    d["ffps_xyz"], d["ffps_feat"] = xyz, feat
    d["ffps_idx_w1"] = oracle.ffps(xyz, feat, 64, 1.0)
    d["ffps_idx_w0"] = oracle.ffps(xyz, feat, 64, 0.0)
    layers = synth.make_mlp_weights([9, 24, 40], rng)
    for i, (w, b) in enumerate(layers):
        d[f"bf16_w{i}"], d[f"bf16_b{i}"] = w, b
    new_xyz = np.ascontiguousarray(xyz[:, :50])
    idx = oracle.ball_query(0.2, 16, xyz, new_xyz)
    fb = oracle.bf16_round(feat)
    d["bf16_idx"] = idx
    d["bf16_pooled"] = oracle.sa_group_mlp_max_bf16(xyz, fb, new_xyz, idx, layers)
    d["bf16_rows"] = oracle.mlp_rows_bf16(np.concatenate([xyz[0, :32], feat[0, :32]], 1), layers)
    boxes = np.zeros((1, 40, 9), np.float32)
    boxes[0, :, 0:2] = rng.uniform(0, 12, (40, 2))
    boxes[0, :, 3:6] = rng.uniform(1.0, 4.0, (40, 3))
    boxes[0, :, 6] = rng.uniform(-3.0, 3.0, 40)
    boxes[0, :, 7] = rng.uniform(0, 1, 40)
    keep, order, count = oracle.nms_bev(boxes, 0.1, 0.2)
    d["nms_boxes"], d["nms_keep"], d["nms_order"], d["nms_count"] = boxes, keep, order, count
    np.savez_compressed(os.path.join(OUT, "extensions.npz"), **d)


if __name__ == "__main__":
    oracle.build()
    config0()
    edge_cases()
    adaptive()
    tiny_detector()
    extensions()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
