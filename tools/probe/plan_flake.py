"""(Diagnosis of a TEST bug, kept as a demonstration: `out.clone()` on the null stream right after `ev.synchronize()` is not ordered
against the next step, which frees `out` and reuses its block on the main stream; FIX=1 waits for the copy.  DESIGN.md §9.)
The scenario of tests/test_gpu_plan.py::test_replayed_steps_equal_eager_steps_on_other_inputs[bf16], repeated, with details on a
mismatch: which step, plan state, how many values differ and where (box index / column), and whether a second eager run of the same
batch agrees with the first (is the EAGER path deterministic?).  Bounded: REPS scenarios in one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sad_amd  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sad_amd import config, synth  # noqa: E402
from sad_amd.detector import SADDetector  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
cfg = config.TINY
w = synth.make_weights(cfg, 0)
batches = [torch.from_numpy(np.ascontiguousarray(synth.make_tiny_batch(10 * k, 3, cfg.n_points))).to(dev) for k in range(5)]
bad = 0
for rep in range(REPS):
    eager = SADDetector(cfg, w, dev, dtype=dtype)
    eager.use_plans = False
    want, want2 = [], []
    for b in batches:
        out, ev = eager.submit(b)
        ev.synchronize()
        want.append(out.clone())
        if os.environ.get("FIX"):
            torch.cuda.current_stream().synchronize()
    for b in batches:                      # the eager path against itself
        out, ev = eager.submit(b)
        ev.synchronize()
        want2.append(out.clone())
        if os.environ.get("FIX"):
            torch.cuda.current_stream().synchronize()
    for k in range(5):
        if not torch.equal(want[k], want2[k]):
            bad += 1
            d = want[k] != want2[k]
            print(f"rep {rep}: EAGER run differs from EAGER run on batch {k}: {int(d.sum())} values, boxes {d.any(-1).nonzero().tolist()[:6]}", flush=True)
    det = SADDetector(cfg, w, dev, dtype=dtype, streams=(eager._sides, eager._mains))
    for i in range(40):
        out, ev = det.submit(batches[i % 5])
        ev.synchronize()
        if not torch.equal(out, want[i % 5]):
            bad += 1
            d = out != want[i % 5]
            idx = d.nonzero()
            print(f"rep {rep}: step {i} (slot {i % det._plan_ring}, plans {det.use_plans}, refused {det.plan_refused}, replays {det.plan_replays}) differs: "
                  f"{int(d.sum())} values in {int(d.any(-1).sum())} boxes, columns {sorted(set(idx[:, 2].tolist()))}, first {idx[:4].tolist()}, "
                  f"max abs diff {float((out - want[i % 5]).abs().max()):.4g}", flush=True)
print(f"{dtype}: {bad} mismatches in {REPS} scenarios", flush=True)
