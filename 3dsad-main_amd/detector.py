"""The measured path end to end: 3-stage SA backbone -> size-adaptive cluster layer -> box/cls head
(SPEC.md §7-§9, topology §11).  The upstream reference (``/root/reference/README.md:1-2``) names the
idea ("size adaptive clustering") and ships no code; the mechanism is this repository's SPEC.md §8.

Stream plan.  Sampling (fps -> gather) depends on coordinates only, so the serial FPS chain of all
three stages runs on its own HIP stream, one 1024-thread workgroup per scene, while the grouping /
MFMA-MLP kernels of the earlier stages fill the rest of the chip on the main stream; events hand
each stage's centroids over.
"""
import ctypes
import math
import os
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import _lib as _libmod
from . import ops
from . import plan as _plan
from ._lib import check, lib
from .config import DetectorConfig
from .sa_module import SAModuleMSG


class SADDetector(nn.Module):
    _streams_created = {}         # device index -> streams made by every detector of this process there (never destroyed: torch pools them)

    def __init__(self, cfg: DetectorConfig, weights: dict, device, overlap_fps: bool = True,
                 n_fps_streams: Optional[int] = None, n_main_streams: int = 2, nested_fps_shortcut: bool = True,
                 dtype: str = "f32", query_on_sampling_stream: bool = True, streams=None, n_extra_streams: int = 2):
        """``dtype="bf16"``: every MLP runs on the bf16 matrix-core path (SPEC.md §14, BASELINE.json
        configs[4]); sampling, ball query and box decode are unchanged.
        ``streams`` = (sampling streams, main streams): reuse these instead of creating new ones — a process that builds
        several detectors should share one set, since every stream ever created keeps its place among the
        ``GPU_MAX_HW_QUEUES`` hardware queues and streams beyond that number share queues (a detector built after 16
        streams exist ran its FPS chains at half speed: measured).  A detector that makes its own streams makes them through
        ``_runtime.placed_streams`` (each main stream alone on its dispatch pipe) together with ``n_extra_streams`` more for the
        caller: ``det.extra_streams[0]`` is meant for the gather (``dist.AsyncBoxGather(dev, stream=...)``), ``[1]`` for the ingest
        stream of ``pipeline.IngestPipeline`` — a stream the caller creates later lands wherever the next queue number falls,
        possibly on a main stream's pipe (4 - 7 % of the pipelined step: DESIGN.md §5).  Detectors of one process that ask for
        the same stream counts get the SAME set (``_runtime.placed_streams``)."""
        super().__init__()
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        self.dtype = dtype
        mlp_cls = ops.PackedMLP if dtype == "f32" else ops.PackedMLPBf16
        self.cfg = cfg
        self.device = torch.device(device)
        c = cfg.in_feat
        self.stages = []
        for si, st in enumerate(cfg.stages):
            w = {f"b{i}": weights[f"sa{si + 1}.b{i}"] for i in range(len(st.mlps))}
            if st.agg:
                w["agg"] = weights[f"sa{si + 1}.agg"]
            m = SAModuleMSG(c, st, self.device, w, name=f"sa{si + 1}", dtype=dtype)
            self.stages.append(m)
            c = m.out_channels
        self.cand_mlp = mlp_cls(weights["cand"], False, self.device,
                                      relu_mask=(1 << (len(weights["cand"]) - 1)) - 1, name="cand")
        self.cluster_branches = [mlp_cls(weights[f"cluster.b{i}"], True, self.device,
                                               name=f"cluster.b{i}")
                                 for i in range(len(cfg.cluster_mlps))]
        self.cluster_cat = sum(m[-1] for m in cfg.cluster_mlps)
        self.cluster_agg = mlp_cls(weights["cluster.agg"], False, self.device,
                                         name="cluster.agg")
        self.head = mlp_cls(weights["head"], False, self.device,
                                  relu_mask=(1 << (len(weights["head"]) - 1)) - 1, name="head")
        # aggregation + head as ONE chain (same fmaf chains, activations stay in LDS, one launch and
        # no HBM round trip of the cluster features); used when no trace is requested
        n_fused = len(weights["cluster.agg"]) + len(weights["head"])
        self.agg_head = None
        if n_fused <= 4 and ((dtype == "f32" and not os.environ.get("SAD_F32_NO_FUSE_HEAD")) or os.environ.get("SAD_BF16_FUSE_HEAD")):
            # (bf16: a measurement switch — the fused chain runs the round-1 tiled kernel)
            self.agg_head = mlp_cls(list(weights["cluster.agg"]) + list(weights["head"]), False, self.device,
                                    relu_mask=(1 << (n_fused - 1)) - 1, name="cluster.agg+head")
        self._anchor = (ctypes.c_float * 3)(*cfg.anchor_car)
        self._anchors = (ctypes.c_float * 9)(*[v for a in cfg.anchors for v in a])
        self.overlap_fps = overlap_fps
        # The SA ball queries need coordinates only: run them right behind the sampling chain, off the
        # main stream (whose MLP kernels they then overlap); the adaptive cluster query stays on main.
        self.query_on_sampling_stream = query_on_sampling_stream
        self.poison_buffers = False        # test knob: fill the (otherwise uninitialised) pooling buffers with NaN
        # FPS of an FPS-ordered point set is the identity prefix: stage s+1 samples the first M_{s+1}
        # centroids of stage s (proof in _sample_stage).  Only stage 1 runs the FPS kernel.
        self.nested_fps_shortcut = nested_fps_shortcut
        # Sampling streams, used round-robin by consecutive calls: one FPS chain keeps only B CUs
        # busy (one workgroup per scene, a serial chain of M steps), so with input_ready=True the
        # chains of several consecutive batches run side by side while the main stream works
        # through the grouping / MLP kernels of earlier batches.
        # (main + sampling streams + the caller's gather stream must not outnumber the hardware queues: _runtime.py)
        from . import HW_QUEUES_STATE, _runtime
        if n_fps_streams is None:                 # default: eight, or every sampling stream of a set that was handed in
            n_fps_streams = 8 if streams is None else len(streams[0])
        n_side, n_main = (max(1, n_fps_streams) if overlap_fps else 0), max(1, n_main_streams)
        if streams is not None:
            if len(streams[0]) < n_side or len(streams[1]) < n_main:
                raise ValueError(f"streams: need {n_side} sampling and {n_main} main streams")
        dkey = self.device.index if self.device.index is not None else torch.cuda.current_device()   # (hardware queues are per device)
        new_set = streams is None and (str(self.device), n_side, n_main, max(0, n_extra_streams)) not in _runtime._SETS
        made = SADDetector._streams_created.get(dkey, 0) + (len(_runtime.placement_order(n_side, n_main, max(0, n_extra_streams))) if new_set else 0)
        SADDetector._streams_created[dkey] = made
        _runtime.check_stream_budget(max(n_side + n_main, made) + 1, HW_QUEUES_STATE)
        # (own streams: created and touched in the order that leaves each main stream alone on its dispatch pipe, _runtime.py)
        own = None if streams is not None else _runtime.placed_streams(self.device, n_side, n_main, max(0, n_extra_streams))
        self.extra_streams = list(streams[2]) if (streams is not None and len(streams) > 2) else (own[2] if own is not None else [])
        self._sides = list(streams[0][:n_side]) if streams is not None else own[0]
        self._calls = 0
        # submit(): consecutive batches alternate between main streams, so the tail of one batch's
        # kernels (few workgroups left, most CUs idle) overlaps the next batch's kernels.
        self._mains = list(streams[1][:n_main]) if streams is not None else own[1]
        self._submits = 0
        # Step plans (plan.py): submit() records the launches of a step once per ring slot and replays them afterwards.  The
        # ring is a multiple of both stream counts, so a slot always meets the same (main, sampling) stream pair, and long
        # enough (>= 12) that a caller with up to 11 steps in flight never waits for a slot; a slot's buffers (~0.3 GB for 32
        # KITTI-shaped scenes) live as long as its plan.  ``use_plans=False`` (or any step that cannot be recorded) = the eager path.
        self.use_plans = not os.environ.get("SAD_NO_PLANS")      # (A/B switch for measurements: the eager path)
        self.plan_refused = None           # why recording was given up, if it was
        self.plan_replays = 0
        self._plans = {}
        period = (len(self._mains) * max(1, len(self._sides))) // math.gcd(len(self._mains), max(1, len(self._sides)))
        self._plan_ring = period * ((12 + period - 1) // period)

    def submit(self, points: torch.Tensor, post=None, ready=None):
        """Throughput entry point: enqueue one batch on the next main stream (round-robin) and
        return (result, done_event) without waiting.  ``points`` must already be resident and not
        be written by queued work (same promise as ``input_ready=True``).  ``post(boxes)`` is called
        with that stream current (e.g. the all_gather of a sharded job).  ``done_event`` covers
        everything that produces ``result``: when the hook moves work to a stream of its own and
        exposes its completion as ``post.event`` (``dist.AsyncBoxGather``), that event is returned,
        otherwise one recorded on the main stream behind the hook.  ``ready``: an event behind whatever produces ``points``
        on another stream (``pipeline.IngestPipeline``: H2D copy + subsample / pad) — the sampling stream and the main stream
        wait for it instead of for each other.
        ``result`` belongs to the main stream that produced it (an eager step: a fresh tensor of that stream's allocator pool; a
        replayed step: the slot's buffer).  A consumer on ANY other stream — the null stream included: torch's streams do not
        synchronise with it — waits for ``done_event`` and then either finishes reading before the next submits reuse the memory
        (``stream.synchronize()`` behind an asynchronous ``clone()``) or calls ``result.record_stream(its_stream)``."""
        st = self._mains[self._submits % len(self._mains)]
        slot = self._submits % self._plan_ring
        self._submits += 1
        self.last_stream = st
        plan = None
        if self._plannable(points):
            key = (slot, tuple(points.shape), ready is not None)
            plan = self._plans.get(key)
            if plan is not None:
                # replay: the slot's buffers are free once the step that last used them has completed
                if plan.done is not None:
                    plan.done.synchronize()
                else:
                    plan.done = torch.cuda.Event()
                if plan.post_done is not None:           # ... and the hook that read the slot's boxes on a stream of its own (the gather)
                    plan.post_done.synchronize()
                    plan.post_done = None
                self._calls += 1
                plan.last_input = points                 # (held until the slot's next use: a replay records no stream use for the allocator)
                out = plan.replay(points.data_ptr(), ready)
                ev = None
                if post is not None:
                    with torch.cuda.stream(st):
                        out = post(out)
                    ev = getattr(post, "event", None)
                    plan.post_done = ev
                plan.done.record(st)
                self.plan_replays += 1
                return out, (ev if ev is not None else plan.done)
        with torch.cuda.stream(st):
            if plan is None and self._plannable(points):
                rec = _plan.Recorder()
                _libmod.set_recorder(rec)
                try:
                    out = self.forward(points, input_ready=True, ready=ready)
                    plan = rec.finish(out, st)
                    self._plans[key] = plan
                except _plan.PlanUnsupported as e:            # something on this step cannot be recorded: stay eager for good
                    _libmod.set_recorder(None)
                    self.use_plans, self.plan_refused = False, str(e)
                    out = self.forward(points, input_ready=True, ready=ready)
                finally:
                    _libmod.set_recorder(None)
            else:
                out = self.forward(points, input_ready=True, ready=ready)
            ev = None
            if post is not None:
                out = post(out)
                ev = getattr(post, "event", None)
            if plan is not None:
                if plan.done is None:
                    plan.done = torch.cuda.Event()
                plan.done.record(st)
                plan.post_done = ev
                if ev is None:
                    ev = plan.done
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(st)
        return out, ev

    def _plannable(self, points) -> bool:
        """May this submit() go through a step plan (plan.py)?  Not while tuning, tracing launches or poisoning buffers, and
        only for the packed float32 input the recorded launches were made for."""
        return (self.use_plans and self.overlap_fps and self.query_on_sampling_stream and not ops.AUTOTUNE
                and ops.LAUNCH_LOG is None and ops.RERUN_LOG is None and not self.poison_buffers
                and points.is_cuda and points.dtype == torch.float32 and points.dim() == 3 and points.is_contiguous())

    def prime_plans(self, points: torch.Tensor) -> int:
        """Record every slot of the step-plan ring now (one eager step each, synchronised at the end) instead of during the
        first steps of serving: a recording step allocates the ~0.3 GB of buffers its slot keeps, fresh from the driver —
        tens of milliseconds that do not belong into a latency-sensitive or timed region (bench.py calls the same steps its
        setup).  Returns the number of steps run (0 when plans are off or ``points`` is not plannable)."""
        if not self._plannable(points):
            return 0
        n = 0
        for _ in range(self._plan_ring):
            self.submit(points)
            n += 1
        torch.cuda.synchronize(self.device)
        return n

    def clear_plans(self) -> None:
        """Drop every recorded step plan and the buffers they own (geometry changes call this)."""
        self._plans.clear()

    def autotune(self, points: torch.Tensor) -> dict:
        """One synchronous forward pass during which every MLP launch times its workgroup
        geometries on the real shapes and keeps the fastest.  Returns {launch name: geometry}."""
        self.clear_plans()
        prev, ops.AUTOTUNE = ops.AUTOTUNE, True
        ov, self.overlap_fps = self.overlap_fps, False
        try:
            self.forward(points)
            torch.cuda.synchronize()
        finally:
            ops.AUTOTUNE, self.overlap_fps = prev, ov
        return self.geometry()

    def mlps(self) -> list:
        """Every packed MLP chain of the detector (branches, aggregations, candidate MLP, head, fused chain)."""
        out = [b for m in self.stages for b in m.branches] + [m.agg for m in self.stages if m.agg]
        out += [self.cand_mlp, self.cluster_agg, self.head] + self.cluster_branches
        if self.agg_head is not None:      # (tuned by autotune's forward pass, which runs the fused chain)
            out.append(self.agg_head)
        return out

    def geometry(self) -> dict:
        """{launch name: workgroup geometry code} of every chain that has one (tuned or loaded)."""
        g = {}
        for m in self.mlps():
            code = list(m._geom.values())[-1] if m._geom else m.default_geometry
            if code:
                g[m.name] = int(code)
        return g

    def set_geometry(self, geometry: dict) -> None:
        """Use a saved ``geometry()`` / ``autotune()`` result instead of measuring again (reproducible
        runs: rocprofv3 then sees steady-state launches only).  Codes are validated by the C-ABI at
        launch time (an unusable code raises, it is never silently replaced)."""
        self.clear_plans()
        for m in self.mlps():
            if m.name in geometry:
                m._geom.clear()
                m.default_geometry = int(geometry[m.name])

    def _sample_stage(self, si: int, cur: torch.Tensor) -> torch.Tensor:
        """Centroids of stage si from the previous stage's centroids (or the scene for si = 0).

        Shortcut for si > 0 (exact, SPEC.md §2).  `cur` is the FPS pick sequence p_0, p_1, ... of the
        previous stage.  Running FPS on it starts at p_0; assume its first k picks are p_0..p_{k-1}.
        Every point's min-distance to that set is the same number in both runs (same coordinates,
        same expression), and p_k maximised it over the WHOLE previous input, hence also over the
        subset.  Ties: the previous run took the tied point with the lowest original index, i.e.
        tied points appear in `cur` in index order, so "lowest position in cur" selects the same
        point.  By induction fps(cur, M) = (0, 1, ..., M-1), so the centroids are cur[:, :M]."""
        m = self.stages[si]
        if si > 0 and self.nested_fps_shortcut and m.stage.npoint <= cur.shape[1]:
            return ops.prefix_rows(cur, m.stage.npoint)          # (a library launch, not a framework copy: recordable, plan.py)
        return m.sample(cur)[1]

    def _sample_chain(self, xyz):
        """All three stages' (new_xyz) — coordinates only."""
        out = []
        cur = xyz
        for si in range(len(self.stages)):
            cur = self._sample_stage(si, cur)
            out.append(cur)
        return out

    def _cluster_tables(self, B: int) -> bool:
        """Do both cluster branches run kernels that take a row-packing table (and prepare their pooling slice themselves)?"""
        cfg = self.cfg
        m3, cin = self.stages[-1].stage.npoint, self.stages[-1].out_channels
        fdt = torch.bfloat16 if (self.dtype == "bf16" and self.stages[-1].agg is not None) else torch.float32
        return all(mlp.wants_prescan(B, m3, cfg.n_cand, s_, self.cluster_cat, cin, feat_dtype=fdt)
                   for mlp, s_ in zip(self.cluster_branches, cfg.cluster_nsamples))

    def _cluster_can_split(self, B: int) -> bool:
        """Split pooling for the cluster layer (sa_module.can_split has the rule for the SA stages)."""
        return (self.dtype == "bf16" and ops.SPLIT_POOL and not ops.AUTOTUNE and self.agg_head is None and self._cluster_tables(B)
                and all(mlp.preferred_geometry == 2 and mlp.out_channels % 16 == 0 for mlp in self.cluster_branches)
                and self.cluster_agg.takes_pooled(B * self.cfg.n_cand, self.cluster_agg.out_channels))

    def forward(self, points: torch.Tensor, trace: Optional[dict] = None,
                input_ready: bool = False, ready=None) -> torch.Tensor:
        """points [B,N,3+in_feat] f32 on the GPU -> boxes [B,K,9].

        ``input_ready=True`` promises that ``points`` is not being produced by work still queued on
        the current stream (e.g. a resident batch): the sampling stream then starts immediately
        instead of waiting for the current stream, so the FPS chain of this call overlaps the MLP
        kernels of the previous call."""
        cfg = self.cfg
        if not points.is_cuda:
            raise RuntimeError("points: expected a GPU tensor (sad_amd has no CPU path)")
        if points.dtype != torch.float32 or points.dim() != 3:
            raise TypeError("points: expected a float32 [B,N,3+C] tensor")
        if not points.is_contiguous():
            ops._unrecordable("points: strided copy")
        points = points.contiguous()
        B, N, D = points.shape
        main = torch.cuda.current_stream()
        rec = _libmod.recorder()                       # a step plan is being recorded (submit): events and waits go into it too
        if rec is not None:
            rec.mark_input(points)

        def new_event(stream):
            if rec is not None:
                return rec.event(stream)
            e = torch.cuda.Event()
            e.record(stream)
            return e

        def wait_for(stream, e):
            if e is None:
                return
            if rec is not None:
                rec.wait(stream, e)
            else:
                stream.wait_event(e)

        def wait_ready(stream):                        # `points` is produced on a third stream: everything below waits for it
            if ready is None:
                return
            if rec is not None:
                rec.wait_ready(stream, ready)
            else:
                stream.wait_event(ready)

        wait_ready(main)
        if self.overlap_fps:
            side = self._sides[self._calls % len(self._sides)]
            self._calls += 1
            wait_ready(side)
            if not input_ready:
                ops._unrecordable("forward without input_ready")
                side.wait_stream(main)
            evs = []
            with torch.cuda.stream(side):
                # coordinates and features as packed operands (two launches of the library: every device operation of a
                # step is one, so a step can be recorded and replayed — plan.py; the intensity channel of a [B,N,4] KITTI
                # batch used to be read in place as a strided view, 2 MB more are written now)
                xyz, feat = ops.split_points(points)
                # the pooling buffers of every stage, one allocation.  With the row-packing scans made here (prescan), the scan
                # zero-fills the few groups the chain kernels combine with an atomic max and the buffers stay uninitialised
                # (217 MB per 32-scene KITTI step not filled); otherwise one zero fill, covered by ev_xyz
                prep = self.query_on_sampling_stream and not ops.AUTOTUNE
                # bf16 mode: split pooling per stage where it applies (sa_module.can_split: bf16 pooled rows + continuation rows,
                # plain stores, nothing to fill; the aggregation layer takes the maximum as it reads) — not when a trace is asked
                # for (it hands out the pooled buffers) or the buffers are poisoned
                split_ok = prep and trace is None and not self.poison_buffers and self.dtype == "bf16"
                n_in = [N] + [m.stage.npoint for m in self.stages[:-1]]
                splits = []
                for si, m in enumerate(self.stages):
                    prev_agg = si > 0 and self.stages[si - 1].agg is not None
                    splits.append(split_ok and m.can_split(B, n_in[si], m.stage.npoint, feat=feat if si == 0 else None,
                                                           feat_dtype=torch.bfloat16 if prev_agg else torch.float32))
                cluster_tables = self._cluster_tables(B)
                splits.append(split_ok and self._cluster_can_split(B))
                shapes = [(B, m.stage.npoint, m.cat_channels) for m in self.stages]
                shapes.append((B, cfg.n_cand, self.cluster_cat))
                if not prep:
                    ops._unrecordable("zero-filled pooling buffers")
                zeros = (ops._empty if prep else torch.zeros)((sum(a * b_ * c_ for (a, b_, c_), sp in zip(shapes, splits) if not sp),),
                                                              dtype=torch.float32, device=points.device)
                if prep and self.poison_buffers:      # (tests: whatever the kernels do not write must not matter)
                    ops._unrecordable("poisoned buffers")
                    zeros.fill_(float("nan"))
                cats, conts, o = [], [], 0
                for si, (shp, sp) in enumerate(zip(shapes, splits)):
                    if sp:
                        if si < len(self.stages):
                            c_, k_ = self.stages[si].split_buffers(B, shp[1], points.device)
                        else:
                            c_ = ops._empty(shp, dtype=torch.bfloat16, device=points.device)
                            k_ = [ops.cont_buffer(B, shp[1], s_, mlp.out_channels, points.device)
                                  for mlp, s_ in zip(self.cluster_branches, cfg.cluster_nsamples)]
                        cats.append(c_)
                        conts.append(k_)
                        continue
                    n_el = shp[0] * shp[1] * shp[2]
                    cats.append(zeros[o:o + n_el].view(shp))
                    conts.append(None)
                    o += n_el
                if prep and not splits[-1]:
                    # the cluster dispatch scans for itself (its query needs the candidates): fine when both branches run table
                    # kernels (the dispatch's own scan prepares `out`), else zero the buffer here
                    if not cluster_tables:
                        ops._unrecordable("zero fill of the cluster pooling buffer")
                        cats[-1].zero_()
                ev_xyz = new_event(side)
                cur = xyz
                centroids = []
                queries = []
                for si in range(len(self.stages)):
                    prev = cur
                    cur = self._sample_stage(si, cur)
                    centroids.append(cur)
                    # (stage 0 reads `feat`; a later stage reads the previous stage's output, which does not exist yet: a fresh
                    # contiguous tensor, bf16 only when a bf16 aggregation layer produced it)
                    prev_agg = si > 0 and self.stages[si - 1].agg is not None
                    queries.append(self.stages[si].query(prev, cur, prescan=not ops.AUTOTUNE, cat=cats[si] if prep else None,
                                                         feat=feat if si == 0 else None,
                                                         feat_dtype=torch.bfloat16 if (self.dtype == "bf16" and prev_agg) else torch.float32,
                                                         conts=conts[si])
                                   if self.query_on_sampling_stream else None)
                    evs.append(new_event(side))
            points.record_stream(side)
            for t in centroids + [xyz] + ([feat] if feat is not None else []):
                t.record_stream(main)
            for q in queries:
                if q is not None:
                    for part in q:
                        for t in part:
                            if t is not None:          # (a branch whose kernel packs for itself has no prescanned table)
                                t.record_stream(main)
            zeros.record_stream(main)
            for c_, k_ in zip(cats, conts):
                if k_ is not None:
                    c_.record_stream(main)
                    for t in k_:
                        t.record_stream(main)
            wait_for(main, ev_xyz)
        else:
            xyz, feat = ops.split_points(points)
            centroids = self._sample_chain(xyz)
            evs = [None] * len(centroids)
            queries = [None] * len(centroids)
            cats = [None] * (len(centroids) + 1)
            conts = [None] * (len(centroids) + 1)
        cur_xyz, cur_feat = xyz, feat
        for si, m in enumerate(self.stages):
            wait_for(main, evs[si])
            new_xyz = centroids[si]
            cur_feat = m.group_and_pool(cur_xyz, cur_feat, new_xyz, query=queries[si], cat=cats[si], conts=conts[si],
                                        keep=None if trace is None else trace.setdefault(f"sa{si + 1}", {}))
            if trace is not None:
                trace[f"sa{si + 1}"].update(new_xyz=new_xyz, out=cur_feat)
            cur_xyz = new_xyz
        # ---- size-adaptive cluster layer (SPEC.md §8) ----------------------------------------
        K = cfg.n_cand
        M3 = cur_xyz.shape[1]
        # (the first K rows of every scene as a packed operand: one library launch, 4 - 8 MB)
        c = self.cand_mlp.rows(cur_feat if K == M3 else ops.prefix_rows(cur_feat, K))     # [B,K,6]
        cand = ops._empty((B, K, 3), dtype=torch.float32, device=points.device)
        rad = ops._empty((B, K), dtype=torch.float32, device=points.device)
        check(lib().sad_candidates_f32(cur_xyz.data_ptr(), c.data_ptr(), B, M3, K, cfg.shift_max,
                                       cfg.r_min, cfg.r_max, self._anchor, cand.data_ptr(),
                                       rad.data_ptr(), main.cuda_stream), "sad_candidates_f32")
        idxs, cnts = ops.ball_query_multi(cfg.cluster_scales, cfg.cluster_nsamples, cur_xyz, cand, rad,
                                          return_counts=True)
        cat = cats[-1]
        ckont = conts[-1]
        if cat is None and trace is None and not ops.AUTOTUNE and self._cluster_can_split(B):     # (the serial path: no buffers were prepared)
            cat = ops._empty((B, K, self.cluster_cat), dtype=torch.bfloat16, device=points.device)
            ckont = [ops.cont_buffer(B, K, s_, mlp.out_channels, points.device) for mlp, s_ in zip(self.cluster_branches, cfg.cluster_nsamples)]
        if cat is None:
            ops._unrecordable("zero-filled cluster pooling buffer")
            cat = torch.zeros((B, K, self.cluster_cat), dtype=torch.float32, device=points.device)
        calls, off = [], 0
        if ckont is not None:
            # split pooling: the tables come from an explicit scan (the same two launches the dispatch would make itself), the aggregation
            # layer needs them beside the continuation rows
            outs, o_ = [], 0
            for mlp, k_ in zip(self.cluster_branches, ckont):
                outs.append((cat, o_, mlp.out_channels, k_))
                o_ += mlp.out_channels
            cwss = ops.rowscan_multi(idxs, cnts, M3, outs)
        for bi, (mlp, idx, cnt) in enumerate(zip(self.cluster_branches, idxs, cnts)):
            calls.append((mlp, cur_xyz, cur_feat, cand, idx, cat, off, cnt) + ((cwss[bi], ckont[bi]) if ckont is not None else ()))
            off += mlp.out_channels
        ops.grouped_multi(calls)
        # ---- head + decode (SPEC.md §9) -------------------------------------------------------
        if ckont is not None:
            pool = [(w_, k_, s_, mlp.out_channels) for w_, k_, s_, mlp in zip(cwss, ckont, cfg.cluster_nsamples, self.cluster_branches)]
            # (cluster features as bf16: the head rounds its input to bf16 on load, so storing them rounded changes no bit and halves their bytes)
            cfeat = self.cluster_agg.rows(cat, out_dtype=torch.bfloat16, pool=pool)
            o = self.head.rows(cfeat)                                    # [B,K,10]
        elif self.agg_head is not None and trace is None:
            cfeat = None
            o = self.agg_head.rows(cat)                                  # [B,K,10]
        else:
            cfeat = (self.cluster_agg.rows(cat, out_dtype=torch.bfloat16) if (self.dtype == "bf16" and trace is None)     # (as above)
                     else self.cluster_agg.rows(cat))
            o = self.head.rows(cfeat)                                    # [B,K,10]
        boxes = ops._empty((B, K, 9), dtype=torch.float32, device=points.device)
        check(lib().sad_decode_boxes_f32(cand.data_ptr(), o.data_ptr(), B, K, self._anchors,
                                         boxes.data_ptr(), main.cuda_stream), "sad_decode_boxes_f32")
        if trace is not None:
            trace["cluster"] = dict(c=c, cand=cand, radius=rad, ball_idx=idxs, cat=cat, cfeat=cfeat,
                                    head=o)
        return boxes
