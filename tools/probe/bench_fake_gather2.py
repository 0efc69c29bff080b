"""Which part of AsyncBoxGather's stream pattern costs the step ~5 %?  Variants (SAD_FAKE2), nothing runs between the events:
 rec      ev.record(main) only, ev returned
 recwait  + comm.wait_event(ev), ev returned
 full     + done.record(comm), done returned                     (= AsyncBoxGather without the collective)
 reuse    full, with one pre-created pair of events per step-plan slot instead of two new events per step
 nodone   full, but submit() gets no event from the hook (the caller synchronises on the main stream's event)
Measurement only.  usage: SAD_FAKE2=rec python tools/probe/bench_fake_gather2.py <bench.py arguments>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sad_amd  # noqa: E402,F401
from sad_amd import dist as sdist  # noqa: E402

MODE = os.environ.get("SAD_FAKE2", "full")


class Gather:
    def __init__(self, device, group=None):
        self.stream = torch.cuda.Stream(device=device)
        self.event = None
        self.pool = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(16)]
        self.n = 0

    def __call__(self, local_boxes):
        cur = torch.cuda.current_stream()
        if MODE == "reuse":
            ev, done = self.pool[self.n % 16]
            self.n += 1
        else:
            ev, done = torch.cuda.Event(), torch.cuda.Event()
        ev.record(cur)
        if MODE == "rec":
            self.event = ev
            return local_boxes
        self.stream.wait_event(ev)
        if MODE == "recwait":
            self.event = ev
            return local_boxes
        done.record(self.stream)
        self.event = None if MODE == "nodone" else done
        return local_boxes

    def wait(self):
        self.stream.synchronize()


sdist.AsyncBoxGather = Gather
import bench  # noqa: E402

bench.main()
