"""Step plans (3dsad-main_amd/plan.py) and the ingest pipeline (3dsad-main_amd/pipeline.py) on the GPU: a replayed step must
produce the bits of the eager step on inputs it was NOT recorded with, and the pipeline's boxes / NMS result must equal the
oracle chain subsample_pad -> detector_forward -> nms_bev on the same staged files.  Parity is against this repository's
spec-oracle (the upstream reference, /root/reference/README.md:1-2, ships no implementation).  (-m gpu)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_copy_rows_matches_strided_views(sad, dev):
    import torch
    from sad_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    pts = torch.randn((3, 1000, 7), generator=g).to(dev)
    xyz, feat = ops.split_points(pts)
    assert torch.equal(xyz, pts[:, :, :3]) and torch.equal(feat, pts[:, :, 3:])
    xyz4, feat1 = ops.split_points(pts[:, :, :4].contiguous())
    assert torch.equal(xyz4, pts[:, :, :3]) and torch.equal(feat1, pts[:, :, 3:4])
    x = torch.randn((4, 512, 256), generator=g).to(dev)
    assert torch.equal(ops.prefix_rows(x, 256), x[:, :256])          # 16-byte path
    y = torch.randn((4, 37, 3), generator=g).to(dev)
    assert torch.equal(ops.prefix_rows(y, 11), y[:, :11])            # word path (rows of 33 words)
    h = torch.randn((2, 64, 128), generator=g).to(dev).to(torch.bfloat16)
    assert torch.equal(ops.prefix_rows(h, 17), h[:, :17])            # 2-byte elements, whole words per row
    with pytest.raises(RuntimeError):
        ops.copy_rows(x, 0, 8, 4, 16, x.new_empty((4, 16)), dst_stride_words=8)     # dst rows would overlap


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_replayed_steps_equal_eager_steps_on_other_inputs(sad, dev, dtype):
    """40 submits over 5 rotating batches: every slot of the ring records on one batch and replays on others.  A framework
    kernel that escaped the recorder (a copy, a fill) would leave a replay with the recording step's data."""
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    cfg = config.TINY
    w = synth.make_weights(cfg, 0)
    batches = [_t(synth.make_tiny_batch(10 * k, 3, cfg.n_points), dev) for k in range(5)]
    eager = SADDetector(cfg, w, dev, dtype=dtype)
    eager.use_plans = False
    want = []
    for b in batches:
        out, ev = eager.submit(b)
        ev.synchronize()
        # (the copy runs on the null stream, which nothing orders against the main stream that owns `out`: it has to be DONE before
        # the next eager step frees `out` and reuses its block — the consumer's side of the submit() contract)
        want.append(out.clone())
        torch.cuda.current_stream().synchronize()
    det = SADDetector(cfg, w, dev, dtype=dtype, streams=(eager._sides, eager._mains))
    for i in range(40):
        out, ev = det.submit(batches[i % 5])
        ev.synchronize()
        assert torch.equal(out, want[i % 5]), f"step {i} (slot {i % det._plan_ring}, {'replay' if i >= det._plan_ring else 'record'}) differs from the eager step"
    assert det.plan_refused is None and det.plan_replays == 40 - det._plan_ring
    assert len(det._plans) == det._plan_ring
    # steps in flight: results of a pipelined run (no waiting between submits) equal the eager ones too
    outs = []
    for i in range(24):
        out, ev = det.submit(batches[(i * 3) % 5])
        outs.append((out.clone() if False else out, ev, (i * 3) % 5))
        if len(outs) > 6:
            o, e, k = outs.pop(0)
            e.synchronize()
            assert torch.equal(o, want[k])
    for o, e, k in outs:
        e.synchronize()
        assert torch.equal(o, want[k])
    # a geometry change drops the plans
    det.set_geometry({})
    assert not det._plans


def test_plan_falls_back_when_a_step_cannot_be_recorded(sad, dev):
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    cfg = config.TINY
    w = synth.make_weights(cfg, 0)
    b = _t(synth.make_tiny_batch(0, 2, cfg.n_points), dev)
    det = SADDetector(cfg, w, dev)
    det.poison_buffers = True                    # a NaN fill is a framework kernel: such steps stay eager
    out, ev = det.submit(b)
    ev.synchronize()
    assert not det._plans and det.plan_refused is None       # (not plannable: never tried)
    det.poison_buffers = False
    ref, ev = det.submit(b)
    ev.synchronize()
    assert torch.equal(out, ref) and len(det._plans) == 1
    assert det.prime_plans(b) == det._plan_ring and len(det._plans) == det._plan_ring and det.plan_replays == 1     # (one slot was recorded already)
    # a strided input view is refused by the recorder, the step runs eagerly and gives the same boxes
    det2 = SADDetector(cfg, w, dev, streams=(det._sides, det._mains))
    wide = torch.zeros((2, cfg.n_points, 6), device=dev)
    wide[:, :, :4] = b
    out2, ev = det2.submit(wide[:, :, :4])
    ev.synchronize()
    assert torch.equal(out2, ref) and not det2._plans


def test_kitti_size_replay_matches_oracle(orc, sad, dev):
    """4 KITTI-shaped scenes through recorded + replayed steps (default geometry): boxes within 1e-4 of the oracle, labels equal."""
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    cfg = config.KITTI
    w = synth.make_weights(cfg, 0)
    pts = [synth.make_batch(50 + 4 * k, 4, cfg.n_points) for k in range(2)]
    det = SADDetector(cfg, w, dev)
    dpts = [_t(p, dev) for p in pts]
    for i in range(det._plan_ring + 2):
        out, ev = det.submit(dpts[i % 2])
    ev.synchronize()
    assert det.plan_replays >= 2
    k = (det._plan_ring + 1) % 2
    want = orc.detector_forward(pts[k], cfg, w, skip_padding=True)
    got = out.cpu().numpy()
    rel = float((np.abs(got.astype(np.float64) - want) / (1.0 + np.abs(want))).max())
    assert rel <= 1e-4 and np.array_equal(got[:, :, 8], want[:, :, 8])


def test_ingest_pipeline_equals_oracle_chain(orc, sad, dev):
    """Ragged files in pinned memory -> H2D -> subsample_pad -> detector -> NMS -> D2H, several steps in flight, against
    oracle.subsample_pad -> detector_forward -> nms_bev on the same staged files (scenes with more, fewer and exactly
    n_points points; an empty scene)."""
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    from sad_amd.pipeline import IngestPipeline
    cfg = config.TINY
    w = synth.make_weights(cfg, 0)
    B = 4
    det = SADDetector(cfg, w, dev)
    pipe = IngestPipeline(det, B, cols=4, max_points_per_scene=2 * cfg.n_points, in_slots=3, out_slots=4, iou_thr=0.3, score_thr=0.2)
    counts = [[3000, 1500, cfg.n_points, 2500], [1000, 4000, 0, 2048], [2200, 2100, 1900, 3900]]
    staged = []
    for k, cs in enumerate(counts):
        scenes = [synth.make_tiny_batch(100 + 10 * k + i, 1, max(n, 1))[0][:n] for i, n in enumerate(cs)]
        assert pipe.stage(k, scenes) == sum(cs) * 16
        staged.append(scenes)
    want = []
    for k in range(3):
        p, o = pipe.staged(k)
        padded = orc.subsample_pad(p, o, cfg.n_points, 0)
        boxes = orc.detector_forward(padded, cfg, w, skip_padding=True)
        want.append((boxes,) + tuple(orc.nms_bev(boxes, 0.3, 0.2)))
    pending = []
    for i in range(30):                                  # plans record and replay underneath; 4 output slots in rotation
        pending.append((pipe.submit(i % 3), i % 3))
        if len(pending) == 3:
            oslot, k = pending.pop(0)
            boxes, order, count = pipe.result(oslot)
            wb, _, wo, wc = want[k]
            rel = float((np.abs(boxes.astype(np.float64) - wb) / (1.0 + np.abs(wb))).max())
            assert rel <= 1e-4, f"step {i}: boxes differ from the oracle chain ({rel:.2e})"
            _, so, sc = orc.nms_bev(np.ascontiguousarray(boxes), 0.3, 0.2)       # the NMS kernel on the device's own boxes: exact
            assert np.array_equal(order, so) and np.array_equal(count, sc)
            assert np.array_equal(count, wc) and np.array_equal(order, wo), f"step {i}: NMS result differs end to end"
            kept = pipe.kept_boxes(boxes, order, count)
            assert [len(x) for x in kept] == list(count)
    assert det.plan_replays > 0 and det.plan_refused is None
    ms = pipe.timed_serial_step(0)
    assert set(ms) == {"h2d", "subsample_pad", "detector_serial", "nms", "d2h"} and all(v >= 0 for v in ms.values())
