"""More variants of the gather's stream pattern without a collective (SAD_FAKE3), and the host time inside submit():
 full      ev.record(main); comm.wait_event(ev); done.record(comm)              (AsyncBoxGather without the collective)
 sidewait  the event is recorded on the step's SAMPLING stream instead of the main stream
 onside    ev.record(main); the step's sampling stream (an existing, busy stream) waits for it, no communication stream
 late      as full, but the wait for step k is enqueued during step k+1
 none      no hook work at all
Measurement only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sad_amd  # noqa: E402,F401
from sad_amd import dist as sdist  # noqa: E402
from sad_amd import detector as sdet  # noqa: E402

MODE = os.environ.get("SAD_FAKE3", "full")
DET = []
HOST = [0.0, 0]

_submit = sdet.SADDetector.submit


def timed_submit(self, *a, **k):
    if not DET:
        DET.append(self)
    t0 = time.perf_counter()
    r = _submit(self, *a, **k)
    HOST[0] += time.perf_counter() - t0
    HOST[1] += 1
    return r


sdet.SADDetector.submit = timed_submit


class Gather:
    def __init__(self, device, group=None):
        # SAD_DUMMY=k: k streams created AND used (a stream takes its hardware queue at first use) in front of the gather stream
        self.dummies = [torch.cuda.Stream(device=device) for _ in range(int(os.environ.get("SAD_DUMMY", "0")))]
        self.stream = torch.cuda.Stream(device=device)
        self.event = None
        self.prev = None
        self.used = False

    def __call__(self, local_boxes):
        if MODE == "none":
            self.event = None
            return local_boxes
        cur = torch.cuda.current_stream()
        if not self.used:
            self.used = True
            for d in self.dummies:
                e = torch.cuda.Event()
                e.record(d)
                d.synchronize()
        det = DET[0]
        side = det._sides[(det._calls - 1) % len(det._sides)]
        ev, done = torch.cuda.Event(), torch.cuda.Event()
        if MODE == "sidewait":
            ev.record(side)
            self.stream.wait_event(ev)
            done.record(self.stream)
            self.event = None
        elif MODE == "onside":
            ev.record(cur)
            side.wait_event(ev)
            self.event = None
        elif MODE == "late":
            ev.record(cur)
            if self.prev is not None:
                self.stream.wait_event(self.prev)
                done.record(self.stream)
            self.prev = ev
            self.event = None
        else:
            ev.record(cur)
            self.stream.wait_event(ev)
            done.record(self.stream)
            self.event = done
        return local_boxes

    def wait(self):
        self.stream.synchronize()


sdist.AsyncBoxGather = Gather
import bench  # noqa: E402

try:
    bench.main()
finally:
    sys.stderr.write(f"HOST submit mean {1e3 * HOST[0] / max(1, HOST[1]):.3f} ms over {HOST[1]} calls\n")
