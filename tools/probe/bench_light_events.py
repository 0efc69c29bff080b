"""bench.py with every untimed torch.cuda.Event replaced by a HIP event created with hipEventDisableSystemFence (no
system-scope release when the event is recorded): what the fences of the step's cross-stream events cost.  Measurement only.
SAD_LIGHT=0 keeps torch's events (control, same wrapper code path); SAD_FAKE2 as in bench_fake_gather2.py selects a gather
pattern without a collective (unset: bench.py's own gather object).
usage: [SAD_BENCH_FORCE_DIST=1] SAD_LIGHT=1 python tools/probe/bench_light_events.py <bench.py arguments>"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hipEventDisableTiming, hipEventDisableSystemFence = 0x2, 0x20000000
FLAGS = hipEventDisableTiming | (hipEventDisableSystemFence if os.environ.get("SAD_LIGHT", "1") == "1" else 0)
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
hip.hipEventQuery.argtypes = [ctypes.c_void_p]
hip.hipEventDestroy.argtypes = [ctypes.c_void_p]

TorchEvent = torch.cuda.Event
torch_wait_event = torch.cuda.Stream.wait_event


class LightEvent:
    def __init__(self):
        self.h = ctypes.c_void_p()
        rc = hip.hipEventCreateWithFlags(ctypes.byref(self.h), FLAGS)
        assert rc == 0, rc

    def record(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream()
        rc = hip.hipEventRecord(self.h, ctypes.c_void_p(s.cuda_stream))
        assert rc == 0, rc

    def wait(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream()
        rc = hip.hipStreamWaitEvent(ctypes.c_void_p(s.cuda_stream), self.h, 0)
        assert rc == 0, rc

    def synchronize(self):
        rc = hip.hipEventSynchronize(self.h)
        assert rc == 0, rc

    def query(self):
        return hip.hipEventQuery(self.h) == 0

    def __del__(self):
        try:
            hip.hipEventDestroy(self.h)
        except Exception:
            pass


def event_factory(*a, **k):
    if a or k.get("enable_timing") or k.get("blocking") or k.get("interprocess"):
        return TorchEvent(*a, **k)
    return LightEvent()


def wait_event(self, ev):
    if isinstance(ev, LightEvent):
        ev.wait(self)
    else:
        torch_wait_event(self, ev)


if os.environ.get("SAD_LIGHT") is not None:
    torch.cuda.Event = event_factory
    torch.cuda.Stream.wait_event = wait_event

import sad_amd  # noqa: E402,F401
from sad_amd import dist as sdist  # noqa: E402

MODE = os.environ.get("SAD_FAKE2")
if MODE:
    class Gather:
        def __init__(self, device, group=None):
            self.stream = torch.cuda.Stream(device=device)
            self.event = None

        def __call__(self, local_boxes):
            cur = torch.cuda.current_stream()
            ev, done = torch.cuda.Event(), torch.cuda.Event()
            ev.record(cur)
            self.stream.wait_event(ev)
            done.record(self.stream)
            self.event = done
            return local_boxes

        def wait(self):
            self.stream.synchronize()

    sdist.AsyncBoxGather = Gather
import bench  # noqa: E402

bench.main()
