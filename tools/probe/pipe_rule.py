"""The rule behind sad_amd._runtime.placed_streams, measured without the detector: a stream of short kernels (A) beside a
stream whose head is a long kernel with a dependent packet behind it (B), B's hardware queue d queue numbers after A's
(d - 1 idle streams touched in between; one process per d so that the numbering starts the same way).  Prints the time of
A's kernels per d: if queues four apart share a dispatch pipe, d = 4 and d = 8 stand out.
usage: python tools/probe/pipe_rule.py            (runs d = 1 .. 8 as child processes)
       python tools/probe/pipe_rule.py <d>        (one measurement)"""
import os
import subprocess
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")      # (before HIP initialises: with the default of 4, streams share whole QUEUES)


def one(d: int) -> None:
    import torch
    dev = torch.device("cuda:0")

    def touch(s):
        e = torch.cuda.Event()
        e.record(s)
        s.synchronize()
        return s

    x = torch.zeros(256, device=dev)
    y = torch.zeros(256, device=dev)
    z = torch.zeros(256, device=dev)
    torch.cuda.synchronize()
    c = touch(torch.cuda.Stream(device=dev))             # holds A back until all of A's kernels are enqueued
    a = touch(torch.cuda.Stream(device=dev))
    keep = [touch(torch.cuda.Stream(device=dev)) for _ in range(d - 1)]
    b = touch(torch.cuda.Stream(device=dev))
    n = 600
    res = {}
    for mode in ("alone", "beside"):
        vals = []
        for rep in range(5):
            torch.cuda.synchronize()
            if mode == "beside":
                with torch.cuda.stream(b):
                    torch.cuda._sleep(120_000_000)       # ~60 ms of one spinning thread
                    y.add_(1.0)                          # a packet that waits for it at the head of B's queue
            with torch.cuda.stream(c):
                torch.cuda._sleep(12_000_000)            # ~6 ms: the host enqueues A's 600 launches meanwhile
                z.add_(1.0)
                gate = torch.cuda.Event()
                gate.record()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(a):
                a.wait_event(gate)
                e0.record()
                for _ in range(n):
                    x.add_(1.0)
                e1.record()
            a.synchronize()
            vals.append(e0.elapsed_time(e1) * 1e3 / n)
            torch.cuda.synchronize()
        vals.sort()
        res[mode] = vals[len(vals) // 2]
    print(f"d={d}: A's queued short kernels {res['alone']:.2f} us each alone, {res['beside']:.2f} us beside a blocked queue {d} numbers later",
          flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(int(sys.argv[1]))
    else:
        for d in range(1, 13):
            subprocess.run([sys.executable, __file__, str(d)], check=False)
