"""Uninitialised reads made visible: before a detector step, every stream's allocator pool is filled with a poison pattern (large
freed blocks of 0x7FC0 7FC0 = NaN as float32 AND as bfloat16 pairs, or of 0xFF bytes), so that any torch.empty of the step hands out
poisoned memory.  The step's boxes must equal those of a run on fresh memory.  usage: python tools/probe/poison_pools.py [cfg] [dtype]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sad_amd  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sad_amd import config, synth  # noqa: E402
from sad_amd.detector import SADDetector  # noqa: E402


def poison(streams, dev, pattern, mb=192):
    """Fill and free `mb` MB in blocks of several sizes on every stream (the caching allocator keeps them per stream)."""
    torch.cuda.synchronize()
    for s in streams:
        with torch.cuda.stream(s):
            keep = []
            for nbytes in (64 << 20, 32 << 20, 16 << 20, 8 << 20, 4 << 20, 2 << 20, 1 << 20, 1 << 20, 512 << 10, 512 << 10, 256 << 10, 64 << 10, 16 << 10):
                for _ in range(2):
                    t = torch.empty((nbytes // 4,), dtype=torch.int32, device=dev)
                    t.fill_(pattern)
                    keep.append(t)
            s.synchronize()
            del keep
    torch.cuda.synchronize()


def main():
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "TINY"
    dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    cfg = getattr(config, cfgname)
    dev = torch.device("cuda:0")
    w = synth.make_weights(cfg, 0)
    B = 3 if cfgname == "TINY" else 2
    mk = synth.make_tiny_batch if cfgname == "TINY" else (lambda seed, b, n: synth.make_batch(seed, b, n))
    pts = [torch.from_numpy(mk(10 * k, B, cfg.n_points)).to(dev) for k in range(3)]
    det = SADDetector(cfg, w, dev, dtype=dtype)
    det.use_plans = False
    streams = det._sides + det._mains + [torch.cuda.current_stream()]
    bad = 0
    for pat_name, pat in (("nan", 0x7FC07FC0), ("ones", -1), ("big", 0x7F7F7F7F)):
        for k, p in enumerate(pts):
            torch.cuda.empty_cache()
            out, ev = det.submit(p)
            ev.synchronize()
            ref = out.clone()
            for trial in range(3):
                poison(streams, dev, pat)
                tr = {}
                out, ev = det.submit(p)
                ev.synchronize()
                same = torch.equal(out, ref)
                nan = bool(torch.isnan(out).any())
                if not same:
                    bad += 1
                    d = (out != ref) & ~(torch.isnan(out) & torch.isnan(ref))
                    print(f"{cfgname} {dtype} pattern {pat_name} batch {k} trial {trial}: DIFFERS ({int(d.sum())} values, NaN in output: {nan})", flush=True)
    print(f"{cfgname} {dtype}: {bad} differing steps", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
