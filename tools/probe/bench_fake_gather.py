"""bench.py with the step's collective replaced by something else on the same stream pattern: separates what the streams and
events of AsyncBoxGather cost from what the RCCL call costs.  Measurement only.
SAD_FAKE = memcpy (device copy, hipMemcpyAsync) | kernel (an elementwise kernel) | events (nothing between the events) |
           main (an elementwise kernel on the main stream, no communication stream at all)
usage: SAD_BENCH_FORCE_DIST=1 SAD_FAKE=kernel python tools/probe/bench_fake_gather.py <bench.py arguments>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sad_amd  # noqa: E402,F401
from sad_amd import dist as sdist  # noqa: E402

MODE = os.environ.get("SAD_FAKE", "memcpy")


def fake(local_boxes, group=None):
    if MODE == "events":
        return local_boxes
    out = torch.empty_like(local_boxes)
    if MODE == "memcpy":
        out.copy_(local_boxes)
    else:
        torch.mul(local_boxes, 1.0, out=out)
    return out


class MainStreamGather:
    event = None

    def __init__(self, device, group=None):
        pass

    def __call__(self, local_boxes):
        return fake(local_boxes)

    def wait(self):
        torch.cuda.synchronize()


sdist.all_gather_boxes = fake
if MODE == "main":
    sdist.AsyncBoxGather = MainStreamGather
import bench  # noqa: E402

bench.main()
