"""Record logic of bench.py that needs no GPU: the tail record of a timed region, the bf16 quality figures, the pipeline
leg's parity record, and what a secondary leg carries into the headline line."""
import numpy as np

import bench


def test_step_tail_flags_long_steps():
    gaps = [1.0] * 95 + [2.5, 3.0, 9.0, 1.1, 0.9]
    t = bench.step_tail(gaps)
    assert t["p50"] == 1.0 and t["max"] == 9.0 and t["min"] == 0.9
    assert t["over_2x_p50"] == 3                       # 2.5, 3.0 and 9.0
    assert abs(t["mean"] - sum(gaps) / len(gaps)) < 1e-3
    assert bench.step_tail([]) == {}
    # a run whose MEAN is 10 % over its median shows it: that is what the round-4 low readings looked like
    assert t["mean"] > 1.09 * t["p50"]


def _boxes(rng, B=2, K=8):
    b = rng.uniform(-1, 1, (B, K, 9)).astype(np.float32)
    b[..., 3:6] = np.abs(b[..., 3:6]) + 1.0
    b[..., 8] = rng.integers(0, 3, (B, K))
    return b


def test_bf16_quality_identical_and_perturbed():
    rng = np.random.default_rng(0)
    f = _boxes(rng)
    idx = [rng.integers(0, 100, (2, 8, 16)).astype(np.int32), rng.integers(0, 100, (2, 8, 32)).astype(np.int32)]
    q = bench.bf16_quality(f, f, idx, idx)
    assert q["label_agreement"] == 1.0 and q["centre_delta_m"]["max"] == 0.0 and q["size_delta_m"]["max"] == 0.0
    assert q["adaptive_index_sets_differ"]["share_of_candidates"] == 0.0 and q["candidates"] == 16
    b = f.copy()
    b[0, 0, 0] += 0.3                                   # one centre moved by 0.3 m along x
    b[0, 1, 3] *= 1.1                                   # one length 10 % larger
    b[1, 2, 8] = (b[1, 2, 8] + 1) % 3                   # one label flipped
    b[1, 3, 6] += 2 * np.pi + 0.01                      # yaw wraps: 0.01 rad, not 6.29
    idx2 = [i.copy() for i in idx]
    idx2[1][1, 7, 5] += 1                               # one slot of one candidate's wide-radius set
    q = bench.bf16_quality(b, f, idx2, idx)
    assert abs(q["centre_delta_m"]["max"] - 0.3) < 1e-5
    assert abs(q["size_delta_rel"]["max"] - 0.1) < 1e-5
    assert q["label_agreement"] == round(15 / 16, 5)
    assert abs(q["yaw_delta_rad"]["max"] - 0.01) < 1e-4
    assert q["adaptive_index_sets_differ"]["share_of_candidates"] == round(1 / 16, 5)
    assert q["adaptive_index_sets_differ"]["per_branch"] == [0.0, round(1 / 16, 5)]


def test_pipeline_parity_record():
    rng = np.random.default_rng(1)
    want = _boxes(rng, 3, 8)

    def nms(b):                                          # a stand-in with the oracle's signature: keep boxes with score > 0, by score
        B, K, _ = b.shape
        order = np.full((B, K), -1, np.int32)
        count = np.zeros((B,), np.int32)
        for i in range(B):
            k = [j for j in np.argsort(-b[i, :, 7], kind="stable") if b[i, j, 7] > 0]
            order[i, :len(k)] = k
            count[i] = len(k)
        return None, order, count

    _, o, c = nms(want)
    got = (np.concatenate([want, want[:1]], 0), np.concatenate([o, o[:1]], 0), np.concatenate([c, c[:1]], 0))   # device ran 4 scenes, oracle 3
    r = bench.pipeline_parity(got, want, nms)
    assert r["ok"] and r["scenes"] == 3 and r["nms_equal_on_device_boxes"] and r["nms_equal_end_to_end"]
    bad = (got[0].copy(), got[1].copy(), got[2])
    bad[1][0, 0], bad[1][0, 1] = bad[1][0, 1], bad[1][0, 0]          # two ranks swapped by the device
    r = bench.pipeline_parity(bad, want, nms)
    assert not r["ok"] and not r["nms_equal_on_device_boxes"]
    off = (got[0] + np.float32(1e-2), got[1], got[2])
    assert not bench.pipeline_parity(off, want, nms)["ok"]


def test_leg_record_carries_tail_and_quality():
    res = {"metric": "m", "value": 1.0, "unit": "scenes/s", "steps": 10, "warmup": 2, "ms_per_step": 1.0, "dtype": "bf16",
           "config": {"workload": "w", "fps_streams": 6, "scenes_per_gpu": 32},
           "step_ms": dict(bench.step_tail([1.0] * 9 + [5.0]), over_2x_p50_at_steps=[9], note="n"),
           "bf16_quality": {"label_agreement": 0.99}}
    leg = bench.leg_record(res)
    assert leg["step_ms"]["over_2x_p50"] == 1 and leg["step_ms"]["max"] == 5.0 and leg["step_ms_p50"] == 1.0
    assert leg["step_ms"]["over_2x_p50_at_steps"] == [9]
    assert leg["bf16_quality"]["label_agreement"] == 0.99


def test_ragged_scenes_are_ragged_and_seeded():
    a = bench.ragged_scenes(5, 4, 2048)
    b = bench.ragged_scenes(5, 4, 2048)
    assert len({s.shape[0] for s in a}) > 1 and all(s.shape[1] == 4 and s.dtype == np.float32 for s in a)
    assert all(int(0.55 * 2048) <= s.shape[0] <= int(1.75 * 2048) for s in a)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_dispatch_bound_picks_the_larger_roofline():
    # a byte-bound bf16 aggregation: 10 GFLOP = 4 us at 2.5 PFLOP/s, 50 MB = 6.25 us at 8 TB/s
    d = bench.dispatch_bound(10e9, 50e6, 0.025, 2500.0)
    assert d["bound"] == "hbm" and d["frac_of_bound"] == 0.25 and d["algorithmic_mb"] == 50.0
    # the f32 cluster layer: 76 GFLOP = 0.483 ms at 157.3 TFLOP/s, far above its bytes
    d = bench.dispatch_bound(76e9, 300e6, 0.86, 157.3)
    assert d["bound"] == "mfma" and abs(d["frac_of_bound"] - 0.5618) < 1e-3
    assert bench.bytes_of("a+b", {"a": 3, "b": 4, "c": 9}) == 7
