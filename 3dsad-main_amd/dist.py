"""Batch sharding across the GPUs of one node (SPEC.md §10).

Scenes are independent, so the path shards with no data-path collective: rank r processes a
contiguous slice of the batch, and ONE ``all_gather`` of the fixed-shape ``boxes [B_loc,K,9]``
(9 216 B per scene — latency-bound, not link-bound) reassembles the result in rank order.  On
ROCm the ``nccl`` backend is RCCL over xGMI; the same code runs on ``gloo`` for CPU tests.
The upstream reference (``/root/reference/README.md:1-2``) has no distributed code to mirror.
"""
import os
from typing import Callable, Tuple

import torch
import torch.distributed as dist


def _collective_needed(group=None) -> bool:
    """A one-rank group has nothing to exchange and skips the collective — unless ``SAD_DIST_FORCE_COLLECTIVE`` is set:
    the rehearsal of the N > 1 path on a one-GPU box (a one-rank RCCL communicator runs the same ``all_gather_into_tensor``
    on the communication stream, so the backend, the stream hand-off and the per-step cost of the call are exercised)."""
    if not dist.is_available() or not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or bool(os.environ.get("SAD_DIST_FORCE_COLLECTIVE"))


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of ``total`` scenes for ``rank``; requires total % world == 0 so
    every rank contributes the same fixed-shape tensor to the all_gather."""
    if total % world != 0:
        raise ValueError(f"batch {total} is not divisible by world size {world}")
    per = total // world
    return rank * per, (rank + 1) * per


def all_gather_boxes(local_boxes: torch.Tensor, group=None) -> torch.Tensor:
    """local [B_loc,K,9] on every rank -> [world*B_loc,K,9] in rank order (one collective)."""
    if not _collective_needed(group):
        return local_boxes
    world = dist.get_world_size(group)
    local_boxes = local_boxes.contiguous()
    out = torch.empty((world * local_boxes.shape[0],) + tuple(local_boxes.shape[1:]),
                      dtype=local_boxes.dtype, device=local_boxes.device)
    dist.all_gather_into_tensor(out, local_boxes, group=group)
    return out


class AsyncBoxGather:
    """``post`` hook for ``SADDetector.submit``: the all_gather of a step's boxes runs on its own
    stream behind an event, so the compute streams never wait for the other ranks — a straggler
    delays only the collective, not the next batch's kernels.  The returned tensor is valid once
    ``.event`` (recorded on the communication stream right behind the collective; ``None`` when no
    collective ran) has completed, or after ``wait()``.  ``SADDetector.submit`` returns that event,
    so ``ev.wait()`` / ``ev.synchronize()`` on the submit result covers the gather too.  The gathered
    tensor is allocated on the communication stream: a consumer on another stream waits for the
    event and calls ``out.record_stream(its_stream)``."""

    def __init__(self, device, group=None, stream=None):
        # (``stream``: one made by ``_runtime.placed_streams`` together with the pipeline's other streams, so that it does not
        # land on a main stream's dispatch pipe)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=device)
        self.group = group
        self.event = None       # completion of the most recent collective (None: nothing in flight)

    def __call__(self, local_boxes: torch.Tensor) -> torch.Tensor:
        if not _collective_needed(self.group):
            self.event = None
            return local_boxes
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            out = all_gather_boxes(local_boxes, self.group)
            done = torch.cuda.Event()
            done.record(self.stream)
        self.event = done
        local_boxes.record_stream(self.stream)
        return out

    def wait(self) -> None:
        self.stream.synchronize()


def run_sharded(forward: Callable[[torch.Tensor], torch.Tensor], points: torch.Tensor,
                group=None) -> torch.Tensor:
    """``points`` [B,N,D] is the GLOBAL batch (same on every rank, or at least this rank's slice
    valid); each rank runs ``forward`` on its slice and the boxes are all-gathered."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(points.shape[0], rank, world)
    return all_gather_boxes(forward(points[lo:hi]), group)
