"""bench.py with only the streams the headline needs (no eight-stream set for the later legs): does the number of mapped
hardware queues move the serial pass (roofline.frac) or the timed region?  Measurement only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def shared_streams(dev, n_side, n_main):
    from sad_amd import _runtime
    from sad_amd.dist import AsyncBoxGather
    st = bench._STREAMS.get(str(dev))
    if st is None:
        side, main, extra = _runtime.placed_streams(dev, n_side, n_main, 2)
        st = bench._STREAMS[str(dev)] = {"side": side, "main": main, "gather": AsyncBoxGather(dev, stream=extra[0]), "ingest": extra[1]}
    return (st["side"][:n_side], st["main"][:n_main]), st["gather"]


bench.shared_streams = shared_streams
bench.main()
