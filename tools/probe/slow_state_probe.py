"""Which part of a run with a TILED pick for cluster.b0 leaves the process slower afterwards?  bf16 pipelined throughput, then an
f32 detector driven in one of several modes for 60 steps, then the bf16 throughput again.  usage: python slow_state_probe.py MODE
MODE: none | good (geometry 3, plans) | good_eager (geometry 3, plans off) | bad (cluster.b0 tiled 24831: eager + zero fill)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import sad_amd, torch, bench
from sad_amd import config, synth, ops
from sad_amd.detector import SADDetector
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
dev = torch.device("cuda:0")
cfg = config.KITTI
(sides, mains), gather = bench.shared_streams(dev, 6, 2)
w = synth.make_weights(cfg, 0)
batches = [torch.from_numpy(synth.make_batch(32 * k, 32, cfg.n_points)).to(dev) for k in range(4)]
def run(det, steps, depth):
    det.prime_plans(batches[0])
    evs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        evs.append(det.submit(batches[i % 4])[1])
        if len(evs) > depth:
            evs.pop(0).synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
late = len(sys.argv) > 2 and sys.argv[2] == "late"      # "late": the bf16 detector is built (and its plans recorded) AFTER the f32 run, as in bench.py
before = float("nan")
if not late:
    bf = SADDetector(cfg, w, dev, n_fps_streams=6, n_main_streams=2, dtype="bf16", streams=(sides, mains))
    before = run(bf, 300, 8)
if mode != "none":
    f32 = SADDetector(cfg, w, dev, n_fps_streams=3, n_main_streams=2, dtype="f32", streams=(sides[:3], mains))
    geo = json.load(open("profiles/r05_geometry.json"))
    if mode == "bad":
        geo["cluster.b0"] = 24831
    f32.autotune(batches[0]) if late else None        # (bench.py tunes first; the tuner's launches are part of what the process has seen)
    f32.set_geometry(geo)
    if mode == "good_eager":
        f32.use_plans = False
    ms = run(f32, 100, 6)
    print(f"  f32 [{mode}] {ms:.3f} ms/step, plans refused: {f32.plan_refused}")
    del f32
    torch.cuda.empty_cache()
if late:
    bf = SADDetector(cfg, w, dev, n_fps_streams=6, n_main_streams=2, dtype="bf16", streams=(sides, mains))
    bf.autotune(batches[0])
after = run(bf, 300, 8)
print(f"mode {mode}: bf16 before {before:.4f} ms/step ({32 / before * 1e3:.0f}/s), after {after:.4f} ({32 / after * 1e3:.0f}/s)", flush=True)
