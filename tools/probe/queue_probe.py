"""Do the pipeline's streams sit on hardware queues of their own IN THIS PROCESS?  Six FPS chains (32 KITTI-shaped scenes each) on
six sampling streams at once take the time of one when every stream has its queue, twice that when two share one.  Then the
quick pipelined bf16 throughput of the same process.  Run it several times: a slow process shows up in the first line."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import sad_amd, torch, bench
from sad_amd import config, synth, ops
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
cfg = config.KITTI
(sides, mains), gather = bench.shared_streams(dev, 6, 2)
x = torch.from_numpy(synth.make_batch(0, 32, cfg.n_points)).to(dev)[:, :, :3].contiguous()
ops.fps(x, 4096); torch.cuda.synchronize()
def conc(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            ops.fps(x, 4096)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
one = min(conc(sides[:1]) for _ in range(3))
six = min(conc(sides) for _ in range(3))
pairs = {}
allst = [("side%d" % i, s) for i, s in enumerate(sides)] + [("main%d" % i, s) for i, s in enumerate(mains)] + [("gather", gather.stream)]
for i in range(len(allst)):
    for j in range(i + 1, len(allst)):
        t = min(conc([allst[i][1], allst[j][1]]) for _ in range(2))
        if t > 1.5 * one:
            pairs[(allst[i][0], allst[j][0])] = round(t, 2)
w = synth.make_weights(cfg, 0)
batches = [torch.from_numpy(synth.make_batch(32 * k, 32, cfg.n_points)).to(dev) for k in range(4)]
det = SADDetector(cfg, w, dev, n_fps_streams=6, n_main_streams=2, dtype="bf16", streams=(sides, mains))
det.prime_plans(batches[0])
evs = []
t0 = time.perf_counter()
for i in range(300):
    evs.append(det.submit(batches[i % 4])[1])
    if len(evs) > 8:
        evs.pop(0).synchronize()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 300 * 1e3
print(f"one FPS chain {one:.2f} ms, six at once {six:.2f} ms; pairs that serialise: {pairs or 'none'}; bf16 pipelined {ms:.4f} ms/step = {32 / ms * 1e3:.0f} scenes/s", flush=True)
