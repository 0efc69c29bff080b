"""Where does the ingest pipeline's step go?  The timed loop of bench.pipeline_leg in four variants on one box:
resident input (the headline's loop), the pipeline without NMS / D2H, without the H2D copies, and complete."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import sad_amd, bench
from sad_amd import config, synth, ops
from sad_amd.detector import SADDetector
from sad_amd.pipeline import IngestPipeline
dev = torch.device("cuda:0")
cfg, B, nb, depth = config.KITTI, 32, 4, 6
w = synth.make_weights(cfg, 0)
streams, _ = bench.shared_streams(dev, 3, 2)
det = SADDetector(cfg, w, dev, n_fps_streams=3, n_main_streams=2, streams=streams)
res = [torch.from_numpy(synth.make_batch(32 * k, B, cfg.n_points)).to(dev) for k in range(nb)]
det.autotune(res[0])
scenes = [bench.ragged_scenes(3200 + 32 * k, B, cfg.n_points) for k in range(nb)]
pipe = IngestPipeline(det, B, max_points_per_scene=int(1.75 * cfg.n_points) + 1, in_slots=nb, out_slots=depth)
for k in range(nb):
    pipe.stage(k, scenes[k])
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200

def loop(submit, consume):
    for i in range(14):
        submit(i)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    t0 = time.perf_counter()
    pend = []
    for i in range(steps):
        pend.append(submit(i))
        marks[i].record(det.last_stream)
        if len(pend) >= depth:
            consume(pend.pop(0))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    gaps = [marks[i].elapsed_time(marks[i + 2]) / 2 for i in range(steps - 2)]
    return dt, bench.step_tail(gaps)

def sub_res(i):
    return det.submit(res[i % nb])[1]
full_ingest, full_post = pipe._ingest, pipe._post
def no_copy_ingest(slot, oslot):
    d_pts, d_off, batch, _ = pipe._dev[oslot]
    with torch.cuda.stream(pipe.ingest):
        ops.subsample_pad(d_pts, d_off, pipe.n_points, pipe.seed, out=batch)
        ev = torch.cuda.Event(); ev.record(pipe.ingest)
    return batch, ev
for name, ing, post in (("resident input (headline loop)", None, None), ("pipeline, no NMS / D2H", full_ingest, lambda o: None),
                        ("pipeline, no H2D copies", no_copy_ingest, full_post), ("pipeline, complete", full_ingest, full_post),
                        ("resident input again", None, None)):
    if ing is None:
        dt, tail = loop(sub_res, lambda ev: ev.synchronize())
    else:
        pipe._ingest, pipe._post = ing, post
        if ing is no_copy_ingest:                       # fill the device staging once
            for o in range(depth):
                pipe._ingest = full_ingest; pipe._ingest(o % nb, o); pipe._ingest = ing
            torch.cuda.synchronize()
        dt, tail = loop(lambda i: pipe.submit(i % nb), lambda o: pipe.result(o))
    print(f"{name:34s} {dt:.3f} ms/step = {B / dt * 1e3:8.0f} scenes/s   {json.dumps(tail)}", flush=True)
