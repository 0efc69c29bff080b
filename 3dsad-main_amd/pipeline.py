"""The rows either side of the detector on ONE step: ragged point files in pinned host
memory -> asynchronous H2D -> ``ops.subsample_pad`` (SPEC.md §17) -> ``SADDetector.submit`` -> ``ops.nms_bev`` (SPEC.md §13)
-> D2H of the boxes, their rank order and the kept count.  The upstream reference (``/root/reference/README.md:1-2``) has no
loader, no NMS and no pipeline to mirror; the file layout is the public KITTI ``.bin`` one (N x 4 float32), see ``io.py``.

Stream plan.  The copy in and the subsample / pad run on an INGEST stream of their own, so a step's input is prepared while
the main streams still work on earlier steps; an event hands the padded batch to the detector's sampling and main streams.
NMS and the three copies out (kernels that write pinned host memory) run on the step's main stream right behind the box decode
(``post`` hook).  Host memory is
pinned and owned by the pipeline: ``in_slots`` input slots (filled by ``stage``) and ``out_slots`` output slots used round-robin;
a slot is reused only after ``result`` (or ``wait``) of the step that used it.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops


class IngestPipeline:
    def __init__(self, det, batch: int, cols: int = 4, max_points_per_scene: int = 1 << 16, in_slots: int = 4, out_slots: int = 8,
                 iou_thr: float = 0.5, score_thr: float = 0.1, seed: int = 0, ingest_stream: Optional[torch.cuda.Stream] = None):
        self.det, self.B, self.cols = det, int(batch), int(cols)
        self.n_points = det.cfg.n_points
        self.K = det.cfg.n_cand
        self.iou_thr, self.score_thr, self.seed = float(iou_thr), float(score_thr), int(seed)
        self.dev = det.device
        # (default: the detector's second extra stream, made with its other streams so that it is not on a main stream's pipe)
        extra = getattr(det, "extra_streams", [])
        self.ingest = ingest_stream if ingest_stream is not None else (extra[1] if len(extra) > 1 else torch.cuda.Stream(device=self.dev))
        cap = self.B * int(max_points_per_scene)
        # pinned input: the scenes' file bytes back to back (as float32 rows) + the B + 1 row offsets
        self._in_pts = [torch.empty((cap, self.cols), dtype=torch.float32).pin_memory() for _ in range(in_slots)]
        self._in_off = [torch.zeros((self.B + 1,), dtype=torch.int32).pin_memory() for _ in range(in_slots)]
        self._in_rows = [0] * in_slots
        # pinned output: boxes [B,K,9], rank order [B,K] (kept indices, -1 padded), kept count [B]
        self._out = [(torch.empty((self.B, self.K, 9), dtype=torch.float32).pin_memory(),
                      torch.empty((self.B, self.K), dtype=torch.int32).pin_memory(),
                      torch.empty((self.B,), dtype=torch.int32).pin_memory()) for _ in range(out_slots)]
        self._out_ev: List[Optional[torch.cuda.Event]] = [None] * out_slots
        self._next_out = 0
        # device staging per output slot (= per step in flight), allocated once: the raw rows and offsets, the padded batch,
        # the NMS outputs and workspace.  Nothing is allocated per step: a buffer handed back to the caching allocator while
        # other streams still use it waits there for their events, and a step that finds no free block pays for a hipMalloc
        # (measured: a few steps in a hundred took 3 ms longer)
        with torch.cuda.device(self.dev):
            self._dev = [(torch.empty((cap, self.cols), dtype=torch.float32, device=self.dev),
                          torch.empty((self.B + 1,), dtype=torch.int32, device=self.dev),
                          torch.empty((self.B, self.n_points, self.cols), dtype=torch.float32, device=self.dev),
                          ops.nms_bev_buffers(self.B, self.K, self.dev)) for _ in range(out_slots)]

    # ---- host side ------------------------------------------------------------------------------------------
    def stage(self, slot: int, scenes: Sequence[np.ndarray]) -> int:
        """Copy B ragged scenes ([N_i, cols] float32 arrays or memory maps: file order kept) into pinned input slot ``slot``.
        Returns the bytes staged.  (What a loader thread does; not part of ``submit``.)"""
        if len(scenes) != self.B:
            raise ValueError(f"expected {self.B} scenes")
        pts, off = self._in_pts[slot].numpy(), self._in_off[slot].numpy()
        o = 0
        off[0] = 0
        for b, s in enumerate(scenes):
            n = int(s.shape[0])
            if s.ndim != 2 or s.shape[1] != self.cols:
                raise ValueError(f"scene {b}: expected [N, {self.cols}] float32")
            if o + n > pts.shape[0]:
                raise ValueError("input slot too small for this batch (max_points_per_scene)")
            pts[o:o + n] = s
            o += n
            off[b + 1] = o
        self._in_rows[slot] = o
        return o * self.cols * 4

    def staged(self, slot: int) -> Tuple[np.ndarray, np.ndarray]:
        """(points [sum N, cols], offsets [B+1]) views of input slot ``slot`` (the oracle's input in the parity check)."""
        return self._in_pts[slot].numpy()[:self._in_rows[slot]], self._in_off[slot].numpy()

    # ---- device side ----------------------------------------------------------------------------------------
    def _ingest(self, slot: int, oslot: int):
        """H2D + subsample / pad on the ingest stream, into the staging buffers of output slot ``oslot`` (free: the step that
        last used them has completed) -> (padded batch [B,n_points,cols], event)."""
        rows = self._in_rows[slot]
        d_pts, d_off, batch, _ = self._dev[oslot]
        with torch.cuda.stream(self.ingest):
            d_pts[:rows].copy_(self._in_pts[slot][:rows], non_blocking=True)
            d_off.copy_(self._in_off[slot], non_blocking=True)
            ops.subsample_pad(d_pts, d_off, self.n_points, self.seed, out=batch)
            ev = torch.cuda.Event()
            ev.record(self.ingest)
        return batch, ev

    def _post(self, oslot: int):
        host_boxes, host_order, host_count = self._out[oslot]
        nms_out = self._dev[oslot][3]

        def post(boxes):
            _, order, count = ops.nms_bev(boxes, self.iou_thr, self.score_thr, out=nms_out)
            ops.copy_to_host(boxes, host_boxes)          # (kernels writing pinned memory, not copy-engine transfers: ops.copy_to_host)
            ops.copy_to_host(order, host_order)
            ops.copy_to_host(count, host_count)
            return boxes
        return post

    def submit(self, slot: int) -> int:
        """Enqueue one step on input slot ``slot``; returns the output slot whose ``result`` holds this step's boxes."""
        oslot = self._next_out
        self._next_out = (self._next_out + 1) % len(self._out)
        if self._out_ev[oslot] is not None:
            self._out_ev[oslot].synchronize()          # (the slot's previous step has left the device)
        batch, ready = self._ingest(slot, oslot)
        _, ev = self.det.submit(batch, post=self._post(oslot), ready=ready)
        self._out_ev[oslot] = ev
        return oslot

    def wait(self, oslot: int) -> None:
        if self._out_ev[oslot] is not None:
            self._out_ev[oslot].synchronize()

    def result(self, oslot: int):
        """(boxes [B,K,9], order [B,K], count [B]) numpy views of output slot ``oslot`` once its step has completed; row
        order[b, :count[b]] of boxes[b] are the kept boxes in rank order."""
        self.wait(oslot)
        b, o, c = self._out[oslot]
        return b.numpy(), o.numpy(), c.numpy()

    @staticmethod
    def kept_boxes(boxes: np.ndarray, order: np.ndarray, count: np.ndarray) -> List[np.ndarray]:
        """Per scene the kept boxes in rank order ([count_b, 9])."""
        return [boxes[b, order[b, :int(count[b])]] for b in range(boxes.shape[0])]

    def timed_serial_step(self, slot: int) -> dict:
        """One step with nothing overlapped (one stream, the detector's sampling chain on it too) and HIP events between the
        stages: ms of H2D, subsample_pad, detector, NMS, D2H.  A measurement helper, never on the throughput path."""
        st = torch.cuda.current_stream()
        rows = self._in_rows[slot]
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        for o in range(len(self._out)):
            self.wait(o)
        d_pts, d_off, batch, nms_out = self._dev[0]
        hb, ho, hc = self._out[0]
        ov, self.det.overlap_fps = self.det.overlap_fps, False
        try:
            marks[0].record(st)
            d_pts[:rows].copy_(self._in_pts[slot][:rows], non_blocking=True)
            d_off.copy_(self._in_off[slot], non_blocking=True)
            marks[1].record(st)
            ops.subsample_pad(d_pts, d_off, self.n_points, self.seed, out=batch)
            marks[2].record(st)
            boxes = self.det(batch)
            marks[3].record(st)
            _, order, count = ops.nms_bev(boxes, self.iou_thr, self.score_thr, out=nms_out)
            marks[4].record(st)
            ops.copy_to_host(boxes, hb)
            ops.copy_to_host(order, ho)
            ops.copy_to_host(count, hc)
            marks[5].record(st)
            st.synchronize()
        finally:
            self.det.overlap_fps = ov
        names = ("h2d", "subsample_pad", "detector_serial", "nms", "d2h")
        return {n: marks[i].elapsed_time(marks[i + 1]) for i, n in enumerate(names)}
