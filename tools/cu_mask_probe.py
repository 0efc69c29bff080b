"""Does partitioning the CUs between the sampling streams (FPS) and the main streams (MLP) help the pipeline?
hipExtStreamCreateWithCUMask streams wrapped as torch ExternalStreams replace the detector's streams."""
import os, sys, json, time, ctypes
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import config, ops, synth
from sad_amd.detector import SADDetector
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value, device=dev)
cfg = config.KITTI
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_batch(0, 32)).to(dev)
det = SADDetector(cfg, w, dev, overlap_fps=True, n_fps_streams=2, n_main_streams=2)
det.set_geometry(json.load(open("profiles/r02_geometry.json")))
ref = det(pts).clone(); torch.cuda.synchronize()
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
    return (time.perf_counter() - t0) / steps * 1e3, out
t, out = run()
print(f"no masks: {t:.3f} ms/step  same boxes {torch.equal(out, ref)}")
ALL = (1 << 256) - 1
# FPS alone on masked streams; the ball queries + scans of a stage on a third (query) stream
orig_query = [m.query for m in det.stages]
qstreams = {}
def install(qmask):
    for si, m in enumerate(det.stages):
        def q(prev, cur, prescan=False, _o=orig_query[si]):
            side = torch.cuda.current_stream()
            key = side.cuda_stream
            if key not in qstreams:
                qstreams[key] = masked_stream(qmask) if qmask is not None else torch.cuda.Stream(device=dev)
            qs = qstreams[key]
            qs.wait_stream(side)
            with torch.cuda.stream(qs):
                r = _o(prev, cur, prescan=prescan)
            side.wait_stream(qs)
            return r
        m.query = q
for nbits in (32, 48, 64):
    samp = (1 << nbits) - 1
    for qmode in ("unmasked", "main CUs"):
        qstreams.clear()
        install(None if qmode == "unmasked" else ALL & ~samp)
        det._sides = [masked_stream(samp) for _ in range(2)]
        det._mains = [masked_stream(ALL & ~samp) for _ in range(2)]
        t, out = run()
        print(f"FPS on {nbits} CUs, queries {qmode}, main streams on the other CUs: {t:.3f} ms/step  same boxes {torch.equal(out, ref)}")
    qstreams.clear()
    install(None)
    det._sides = [masked_stream(samp) for _ in range(2)]
    det._mains = [torch.cuda.Stream(device=dev) for _ in range(2)]
    t, out = run()
    print(f"FPS on {nbits} CUs, queries unmasked, main streams unmasked: {t:.3f} ms/step  same boxes {torch.equal(out, ref)}")
