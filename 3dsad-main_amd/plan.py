"""Step plans: record the launches of one ``SADDetector`` step once, replay them afterwards.

Why.  A step is ~45 launches of ``libsad_amd.so`` on three streams.  Enqueued through the operator surface every step
pays for 48 ``torch.empty``, ~80 argument checks, the ctypes argument blocks and a dozen stream / event objects: 0.74 ms of
host time on a bf16 step whose device side takes 0.88 ms — the host was next in line to bind.  All of
that is the same from step to step: shapes, geometry codes, streams and (once the buffers are kept) every pointer but the
input's.  So the first step that goes through a ring slot runs the ordinary Python path with a recorder attached — every
C-ABI launch (function, argument tuple), every event record / wait, every buffer — and later steps in that slot replay the
list: one ctypes call per launch, nothing allocated, nothing checked again.  The launches, their order, their streams and
the events between them are exactly the eager ones; only the host work in front of them is gone.  (HIP graphs were measured
in round 3 — ``tools/graph_probe.py`` — and lose overlap: a graph ends with its main branch.)

What makes it sound.
* Every device operation of a step is a launch of the library (strided views become packed operands through
  ``sad_copy_rows_u32``, not through framework copies); a framework kernel on the path cannot be recorded, so the sites
  that could issue one call ``unrecordable()`` — the step then stays eager (``PlanUnsupported``).
* A plan owns every tensor its step allocated (a slot's buffers are never freed or reused by another slot), so recorded
  pointers stay valid; the only pointer that changes between steps is the input's, found by address range when the plan
  is finished and patched on replay.
* A slot is reused only after the step that last used it has completed (``done`` event), so the ring bounds the steps in
  flight by its own length even when the caller does not.
No reference counterpart exists (``/root/reference/README.md:1-2`` is the whole upstream repository).
"""
import ctypes
from typing import List, Optional

import torch

from . import _lib

OP_CALL, OP_RECORD, OP_WAIT, OP_WAIT_READY = 0, 1, 2, 3


class PlanUnsupported(RuntimeError):
    """Raised (while recording only) by a code path that would launch something the recorder cannot see."""


def unrecordable(what: str) -> None:
    """Call in front of any framework kernel on the step (a fill, a strided copy): an error while a plan is recorded,
    nothing otherwise."""
    if _lib.recorder() is not None:
        raise PlanUnsupported(what)


class _RecLib:
    """Stands in for the ctypes library while a step is recorded: launches are passed through AND appended to the plan."""

    def __init__(self, real, rec):
        self._real, self._rec, self._wrapped = real, rec, {}

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not _lib.is_launch(name):
            return fn
        w = self._wrapped.get(name)
        if w is None:
            ops = self._rec.ops

            def w(*args, _fn=fn):
                rc = _fn(*args)
                if rc == 0:
                    ops.append([OP_CALL, _fn, args])
                return rc
            self._wrapped[name] = w
        return w


class Recorder:
    def __init__(self, real_lib=None):
        self.ops: List[list] = []
        self.keep: List[torch.Tensor] = []        # every buffer the step allocated
        self.inputs = []                          # (base pointer, bytes) of the caller's input tensors
        self.lib = _RecLib(_lib.lib() if real_lib is None else real_lib, self)

    # -- called from the host code of the step while it records --------------------------------------------
    def empty(self, *a, **k) -> torch.Tensor:
        t = torch.empty(*a, **k)
        self.keep.append(t)
        return t

    def event(self, stream) -> torch.cuda.Event:
        ev = torch.cuda.Event()
        ev.record(stream)
        self.ops.append([OP_RECORD, ev, stream])
        return ev

    def wait(self, stream, ev) -> None:
        stream.wait_event(ev)
        self.ops.append([OP_WAIT, stream, ev])

    def wait_ready(self, stream, ready) -> None:
        stream.wait_event(ready)
        self.ops.append([OP_WAIT_READY, stream, None])

    def mark_input(self, t: torch.Tensor) -> None:
        self.inputs.append((t.data_ptr(), t.numel() * t.element_size()))

    def finish(self, out: torch.Tensor, stream) -> "StepPlan":
        if len(self.inputs) != 1:
            raise PlanUnsupported("a plan has exactly one input tensor")
        base, nbytes = self.inputs[0]
        patches = []
        for oi, op in enumerate(self.ops):
            if op[0] != OP_CALL:
                continue
            for ai, a in enumerate(op[2]):
                if isinstance(a, int) and not isinstance(a, bool) and base <= a < base + nbytes:
                    patches.append((oi, ai, a - base))
                obj = getattr(a, "_obj", None)               # ctypes.byref(struct): a pointer INSIDE an argument block cannot be patched
                for s in ([obj] if isinstance(obj, ctypes.Structure) else []):
                    for fname, ftype in s._fields_:
                        v = getattr(s, fname)
                        if ftype is ctypes.c_void_p and isinstance(v, int) and base <= v < base + nbytes:
                            raise PlanUnsupported(f"argument block field {fname} points into the input tensor")
        if not patches:
            raise PlanUnsupported("no launch reads the input tensor")
        return StepPlan(self.ops, self.keep, patches, out, stream)


class StepPlan:
    """The recorded launches of one step on one ring slot."""
    __slots__ = ("ops", "keep", "patches", "out", "stream", "done", "post_done", "n_calls", "last_input")

    def __init__(self, ops, keep, patches, out, stream):
        self.ops, self.keep, self.patches, self.out, self.stream = ops, keep, patches, out, stream
        self.done = None                          # completion of the slot's most recent step (event made and recorded by the detector)
        self.post_done = None                     # completion of that step's post hook when it ran on a stream of its own (dist.AsyncBoxGather)
        self.n_calls = sum(1 for o in ops if o[0] == OP_CALL)
        self.last_input = None                    # the input tensor of the slot's most recent step (kept alive until the next one)

    def replay(self, in_ptr: int, ready: Optional[torch.cuda.Event] = None) -> torch.Tensor:
        ops = self.ops
        for oi, ai, off in self.patches:
            a = list(ops[oi][2])
            a[ai] = in_ptr + off
            ops[oi][2] = tuple(a)
        for op in ops:
            k = op[0]
            if k == OP_CALL:
                if op[1](*op[2]):
                    _lib.check(-1, "plan replay")
            elif k == OP_RECORD:
                op[1].record(op[2])
            elif k == OP_WAIT:
                op[1].wait_event(op[2])
            elif ready is not None:
                op[1].wait_event(ready)
        return self.out
