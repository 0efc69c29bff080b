"""Does the FPS kernel keep its single-stream time when several chains run side by side on streams of their own?
    python tools/fps_concurrency.py [N] [M] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import sad_amd
from sad_amd import ops, synth
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
make = synth.make_nuscenes_batch if N > 16384 else synth.make_batch
pts = torch.from_numpy(make(0, B, N)).to(dev)
xyz = pts[:, :, :3].contiguous()
ops.fps(xyz, M); torch.cuda.synchronize()
ALL = [torch.cuda.Stream() for _ in range(8)]
for s in ALL:                                   # warm every stream's allocator pool
    with torch.cuda.stream(s):
        ops.fps(xyz, M)
torch.cuda.synchronize()
for ns in (1, 2, 3, 4, 6, 8):
    streams = ALL[:ns]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(2):
        for s in streams:
            with torch.cuda.stream(s):
                ops.fps(xyz, M)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"N={N} M={M} B={B}: {ns} streams x 2 calls: {dt:.1f} ms total = {dt / (2 * ns):.2f} ms per call (single-stream serial would be {dt / 2:.1f} per round)")
