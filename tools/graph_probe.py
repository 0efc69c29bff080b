"""Can one pipelined step (SADDetector.submit: sampling stream + main stream, events between them) be captured into a HIP graph
and replayed, and is the replayed pipeline faster than the eager one?   usage: python tools/graph_probe.py [n graphs]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=True, n_fps_streams=3, n_main_streams=2)
det.autotune(pts)
for _ in range(6):
    ref = det.submit(pts)[0]
torch.cuda.synchronize()
ref = ref.clone()
def eager(steps=100, depth=6):
    evs = []
    t0 = time.perf_counter()
    for _ in range(steps):
        out, ev = det.submit(pts)
        evs.append(ev)
        if len(evs) > depth:
            evs.pop(0).synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
print(f"eager pipeline: {eager():.3f} ms/step", flush=True)
# capture NG graphs, each one full step on its own capture stream (the detector's side streams fork from it and join back)
graphs, outs = [], []
cap = torch.cuda.Stream(device=dev)
orig_submit_streams = det._mains
try:
    for gi in range(NG):
        g = torch.cuda.CUDAGraph()
        det._mains = [cap]                      # the captured step runs "on" the capture stream
        with torch.cuda.graph(g, stream=cap):
            out = det.forward(pts, input_ready=False)   # (the sampling stream forks from the capture stream)
        graphs.append(g); outs.append(out)
        print(f"captured graph {gi}", flush=True)
finally:
    det._mains = orig_submit_streams
torch.cuda.synchronize()
def replay(ns, steps=96):
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % ns]):        # graph k always on stream k % ns: never two replays of one graph at once
            graphs[i % NG].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
replay(2, 8)
print(f"same boxes after replay: {all(torch.equal(o, ref) for o in outs)}", flush=True)
for ns in (1, 2, 3, NG):
    if NG % ns == 0:
        print(f"graph replay ({NG} graphs on {ns} streams): {replay(ns):.3f} ms/step", flush=True)
