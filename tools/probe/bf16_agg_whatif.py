"""What do the aggregation layers of the bf16 path cost the pipelined step?  The step as it is and with the four aggregation launches not
launched at all (cached outputs: wrong results, a ceiling), alternating on one box; then with only the first three / only cluster.agg cached."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import sad_amd, torch
from sad_amd import config, synth
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
cfg = config.NUSCENES if len(sys.argv) > 1 and sys.argv[1] == "nuscenes" else config.KITTI
make = synth.make_nuscenes_batch if cfg is config.NUSCENES else synth.make_batch
w = synth.make_weights(cfg, 0)
batches = [torch.from_numpy(make(32 * k, 32, cfg.n_points)).to(dev) for k in range(2 if cfg is config.NUSCENES else 4)]
det = SADDetector(cfg, w, dev, n_fps_streams=6, n_main_streams=2, dtype="bf16")
det.autotune(batches[0])
mlps = {f"sa{i + 1}.agg": m.agg for i, m in enumerate(det.stages)}
mlps["cluster.agg"] = det.cluster_agg
real = {k: m.rows for k, m in mlps.items()}
cache = {}
def cached(name, fn):
    def f(x, *a, **k):
        if name not in cache:
            cache[name] = fn(x, *a, **k)
        return cache[name]
    return f
def run(steps, depth=8):
    det.clear_plans()
    det.prime_plans(batches[0])
    evs = []
    t0 = time.perf_counter()
    for i in range(steps):
        evs.append(det.submit(batches[i % len(batches)])[1])
        if len(evs) > depth:
            evs.pop(0).synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
steps = 100 if cfg is config.NUSCENES else 400
for r in range(2):
    for k, m in mlps.items(): m.rows = real[k]
    a = run(steps)
    res = []
    for which in (list(mlps), ["sa1.agg"], ["cluster.agg"]):
        cache.clear()
        for k, m in mlps.items(): m.rows = cached(k, real[k]) if k in which else real[k]
        det.clear_plans(); det.forward(batches[0]); torch.cuda.synchronize()
        b = run(steps)
        res.append(f"without {'+'.join(which) if len(which) < 4 else 'all four'}: {b:.4f} ({100 * (b - a) / a:+.2f} %)")
    print(f"round {r}: as is {a:.4f} ms/step; " + "; ".join(res), flush=True)
