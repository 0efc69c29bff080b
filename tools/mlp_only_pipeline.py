"""What does the sampling side (FPS, SA ball queries, row-packing scans) cost the pipeline?  Runs the timed loop of
bench.py (submit() on two main + two sampling streams) as is, and again with the sampling products of the batch
cached (no FPS / query / scan launches: the sampling streams only zero the pooling buffers)."""
import os, sys, json, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # as bench.py: one hardware queue per stream
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=True, n_fps_streams=2, n_main_streams=2)
det.set_geometry(json.load(open("profiles/r02_geometry.json")))
def run(steps=300, depth=6):
    evs = []
    for _ in range(10):
        det.submit(pts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out, ev = det.submit(pts)
        evs.append(ev)
        if len(evs) > depth:
            evs.pop(0).synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
print(f"full pipeline:            {run():.3f} ms/step")
cache = {}
orig_sample = det._sample_stage
orig_query = [m.query for m in det.stages]
def patch(sample_cached, query_cached):
    cache.clear()
    def sample(si, cur):
        if si not in cache:
            cache[si] = orig_sample(si, cur)
        return cache[si]
    det._sample_stage = sample if sample_cached else orig_sample
    for si, m in enumerate(det.stages):
        def q(prev, cur, prescan=False, _o=orig_query[si], _k=("q", si)):
            if _k not in cache:
                cache[_k] = _o(prev, cur, prescan=prescan)
            return cache[_k]
        m.query = q if query_cached else orig_query[si]
    det.submit(pts)
    torch.cuda.synchronize()      # the cache is filled and complete before any other stream reads it
for sc, qc in ((True, True), (True, False), (False, True)):
    patch(sc, qc)
    print(f"FPS {'cached' if sc else 'run   '}  queries+scans {'cached' if qc else 'run   '}: {run():.3f} ms/step")
