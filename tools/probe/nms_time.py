"""NMS alone on the boxes of one detector step (32 KITTI-shaped scenes): ms per call, both variants.  Under rocprofv3
--kernel-trace --stats the per-kernel split (prep / mask / walk)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sad_amd, torch
from sad_amd import config, synth, ops
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
cfg = config.KITTI
det = SADDetector(cfg, synth.make_weights(cfg, 0), dev)
boxes = det(torch.from_numpy(synth.make_batch(0, 32, cfg.n_points)).to(dev)).clone()
torch.cuda.synchronize()
buf = ops.nms_bev_buffers(32, cfg.n_cand, dev)
for thr, sthr in ((0.5, 0.1), (0.1, 0.1)):
    for single in (False, True):
        ops.nms_bev(boxes, thr, sthr, single_kernel=single, out=buf); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            k, o, c = ops.nms_bev(boxes, thr, sthr, single_kernel=single, out=buf)
        e1.record(); torch.cuda.synchronize()
        print(f"iou_thr {thr} score_thr {sthr} {'single kernel' if single else 'three kernels'}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call, kept {c.float().mean().item():.1f} of {cfg.n_cand}", flush=True)
