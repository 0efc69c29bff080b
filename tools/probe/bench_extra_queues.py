"""bench.py with SAD_EXTRA_QUEUES=k further streams touched after the pipeline's own set (k more hardware queues mapped, all idle):
where does the device stop serving the queues at full speed?  Measurement only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

_shared = bench.shared_streams
KEEP = []


def shared_streams(dev, n_side, n_main):
    import torch
    first = bench._STREAMS.get(str(dev)) is None
    r = _shared(dev, n_side, n_main)
    if first:
        for _ in range(int(os.environ.get("SAD_EXTRA_QUEUES", "0"))):
            s = torch.cuda.Stream(device=dev)
            e = torch.cuda.Event()
            e.record(s)
            s.synchronize()
            KEEP.append(s)
    return r


bench.shared_streams = shared_streams
bench.main()
