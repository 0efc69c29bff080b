"""Every buffer a detector step allocates (ops._empty, torch.empty inside sad_amd) is filled with a pattern first: a step that reads what it
did not write gives different boxes for different patterns.  Eager steps, TINY (default) or another config.
usage: python tools/probe/fill_empty.py [cfg] [dtype]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sad_amd  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sad_amd import config, synth, ops  # noqa: E402
from sad_amd.detector import SADDetector  # noqa: E402

PATTERN = [None]
LOG = []
_real_empty = torch.empty


def filled_empty(*a, **k):
    t = _real_empty(*a, **k)
    if PATTERN[0] is not None and t.is_cuda and t.numel():
        v = t.view(torch.uint8) if t.dtype != torch.uint8 else t
        v.fill_(PATTERN[0])
        LOG.append((tuple(t.shape), str(t.dtype)))
    return t


cfgname = sys.argv[1] if len(sys.argv) > 1 else "TINY"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = getattr(config, cfgname)
dev = torch.device("cuda:0")
w = synth.make_weights(cfg, 0)
B = 3 if cfgname == "TINY" else 2
mk = synth.make_tiny_batch if cfgname == "TINY" else synth.make_batch
pts = [torch.from_numpy(np.ascontiguousarray(mk(10 * k, B, cfg.n_points))).to(dev) for k in range(3)]
det = SADDetector(cfg, w, dev, dtype=dtype)
det.use_plans = False
torch.empty = filled_empty            # (sad_amd calls torch.empty through the module attribute)
res = {}
for name, pat in (("zero", 0x00), ("ff", 0xFF), ("7f", 0x7F), ("nan-ish c0/7f", 0x7F)):
    PATTERN[0] = pat
    outs = []
    for p in pts:
        del LOG[:]
        out, ev = det.submit(p)
        ev.synchronize()
        outs.append(out.clone())
    res[name] = outs
PATTERN[0] = None
ref = res["zero"]
bad = 0
for name, outs in res.items():
    for k, o in enumerate(outs):
        same = torch.equal(o, ref[k]) or bool(((o == ref[k]) | (torch.isnan(o) & torch.isnan(ref[k]))).all())
        if not same:
            bad += 1
            d = o != ref[k]
            print(f"{cfgname} {dtype}: pattern {name} batch {k}: {int(d.sum())} values differ from the zero-filled run, NaN: {bool(torch.isnan(o).any())}", flush=True)
print(f"{cfgname} {dtype}: {bad} differing outputs; buffers filled in the last step: {len(LOG)}", flush=True)
