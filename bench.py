#!/usr/bin/env python3
"""bench.py — scenes/sec of the SA + size-adaptive-cluster + head path on N MI355X (one node).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric / configs[1..3]): per GPU a resident batch of 32 synthetic
KITTI-shaped 16 384-point scenes -> 3-stage multi-radius SA backbone (fp32) -> size-adaptive
cluster layer -> box/cls head -> boxes[32,256,9]; with N > 1 every rank processes its own 32 scenes
(weak scaling) and ONE RCCL all_gather of the boxes closes each step.  A step = one such pass.
Inputs are already in HBM when the timed region starts.

One JSON line on rank 0: the contract fields plus
  roofline     — the dominant kernel (mlp_chain_kernel: every fused gather+MLP+max / MLP launch of a
                 step) against the dense f32 MFMA peak, timed with HIP events on its own stream
                 inside the timed region;
  kernels      — the same for fps / ball_query (against the HBM roofline, algorithmic bytes);
  cpu_baseline — this repository's CPU spec-oracle (kind "port": the upstream reference ships no
                 CPU path) on a bounded sample of the same scenes, same host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP multiplexes streams onto 4 hardware queues by default; this pipeline uses the main stream,
# three sampling streams and two branch streams, and a 7 ms FPS kernel sharing a queue with MLP
# launches would serialise them.  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

PEAK_MFMA_F32_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 MFMA
PEAK_MFMA_BF16_TFLOPS = 2500.0 # same guide: dense bf16 MFMA (not the 2:1-sparsity headline)
PEAK_HBM_GBPS = 8000.0         # HBM3E spec


def usable_cores() -> int:
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, weights, scenes: int, repeats: int = 2):
    """Oracle forward on `scenes` scenes of the same workload; best of `repeats`."""
    import oracle
    from sad_amd import synth
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    oracle.build()
    pts = synth.make_batch(0, scenes, cfg.n_points)
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        oracle.detector_forward(pts, cfg, weights)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": round(scenes / best, 4), "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": f"{scenes} scenes of the same 16384-pt batch through oracle.detector_forward "
                      f"(C + OpenMP, {cores} threads), best of {repeats}; the upstream reference has no CPU path"}


def executed_flops(det, points, cfg, dense=False):
    """Flops the MLP kernels execute on this batch: per grouped launch only the leading rows of each
    group up to the last sample that differs from the first (the kernel's own rule); plain launches
    (aggregation, candidate MLP, head) execute every row."""
    import torch
    from sad_amd import config as _c
    tr = {}
    det.overlap_fps, ov = False, det.overlap_fps
    try:
        det(points, trace=tr)
        torch.cuda.synchronize()
    finally:
        det.overlap_fps = ov
    B = points.shape[0]
    dims = dict(_c.mlp_layers(cfg))

    def chain(d):
        return 2 * sum(a * b for a, b in zip(d[:-1], d[1:]))

    def rows_of(idx):
        if dense:                 # the bf16 path computes every row of every group
            return idx.numel()
        diff = idx != idx[..., :1]
        pos = torch.arange(1, idx.shape[-1] + 1, device=idx.device)
        return int(torch.clamp((diff * pos).amax(-1), min=1).sum().item())

    ex, dense_rows, exec_rows = 0, 0, 0
    per = {}
    for si, st in enumerate(cfg.stages):
        name = f"sa{si + 1}"
        for bi, idx in enumerate(tr[name]["ball_idx"]):
            r = rows_of(idx)
            per[f"{name}.b{bi}"] = r * chain(dims[f"{name}.b{bi}"])
            exec_rows += r
            dense_rows += idx.numel()
        if st.agg:
            per[f"{name}.agg"] = B * st.npoint * chain(dims[f"{name}.agg"])
    for bi, idx in enumerate(tr["cluster"]["ball_idx"]):
        r = rows_of(idx)
        per[f"cluster.b{bi}"] = r * chain(dims[f"cluster.b{bi}"])
        exec_rows += r
        dense_rows += idx.numel()
    K = cfg.n_cand
    for n in ("cand", "cluster.agg", "head"):
        per[n] = B * K * chain(dims[n])
    per["cluster.agg+head"] = 0      # the fused chain is named "cluster.agg+head": fl() sums its parts
    ex = sum(per.values())
    return ex, exec_rows / max(1, dense_rows), per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32, help="scenes per GPU per step")
    ap.add_argument("--no-overlap", action="store_true", help="run FPS on the main stream")
    ap.add_argument("--fps-streams", type=int, default=4, help="sampling streams used round-robin")
    ap.add_argument("--main-streams", type=int, default=2, help="main streams used round-robin by consecutive steps")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-scenes", type=int, default=32)
    ap.add_argument("--no-launch-timing", action="store_true")
    ap.add_argument("--no-autotune", action="store_true", help="use the built-in geometry heuristic")
    ap.add_argument("--config", choices=("kitti", "nuscenes"), default="kitti",
                    help="nuscenes = BASELINE configs[4] shape (65536-pt scenes, 4 extra channels); secondary, "
                         "use with --dtype bf16 --batch 8 --no-cpu; the headline metric is the kitti default")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="bf16 = SPEC.md 14 mode (configs[4]): MLPs on the bf16 matrix cores, dense rows; "
                         "the headline metric is the f32 default")
    ap.add_argument("--opt", action="append", default=[], help="tuning knob key=value (sad_set_option)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import sad_amd  # noqa: F401
    from sad_amd import config, ops, synth
    from sad_amd.detector import SADDetector
    from sad_amd.dist import AsyncBoxGather

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only path (no CPU fallback)")
    if os.environ.get("SAD_BENCH_ONE_DEVICE"):      # rehearsal of the N>1 path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SAD_BENCH_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:                                                        # rehearsal on a one-GPU box
            dist.init_process_group(backend)

    from sad_amd import _lib
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    cfg = config.KITTI if args.config == "kitti" else config.NUSCENES
    weights = synth.make_weights(cfg, 0)
    det = SADDetector(cfg, weights, dev, overlap_fps=not args.no_overlap, n_fps_streams=args.fps_streams,
                      n_main_streams=args.main_streams, dtype=args.dtype)
    PEAK = PEAK_MFMA_F32_TFLOPS if args.dtype == "f32" else PEAK_MFMA_BF16_TFLOPS
    gather = AsyncBoxGather(dev)       # the step's one collective, off the compute streams
    B = args.batch
    make = synth.make_batch if args.config == "kitti" else synth.make_nuscenes_batch
    points = torch.from_numpy(make(rank * B, B, cfg.n_points)).to(dev)
    torch.cuda.synchronize()

    if os.environ.get("SAD_NO_MERGE_BF16"):
        ops.MERGE_BF16 = False
    tuned = None if args.no_autotune else det.autotune(points)

    def step():
        out, _ = det.submit(points, post=gather)
        return out

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    # per-launch HIP events are recorded on every 8th step of the timed region (a sample: event
    # packets between kernels cost a few per cent of throughput when every launch carries them)
    log = None if args.no_launch_timing else []
    timed_steps = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        sample = log is not None and i % 8 == 4
        ops.LAUNCH_LOG = log if sample else None
        timed_steps += int(sample)
        out = step()
    ops.LAUNCH_LOG = None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert out.shape == (world * B, cfg.n_cand, 9) and bool(torch.isfinite(out).all())
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        work = config.work_per_scene(cfg)
        per_kind = {}
        per_name = {}
        for kind, name, e0, e1 in (log or []):
            ms = e0.elapsed_time(e1)
            per_kind[kind] = per_kind.get(kind, 0.0) + ms
            per_name[(kind, name)] = per_name.get((kind, name), 0.0) + ms
        steps = args.steps
        res = {
            "metric": ("scenes/sec (16384-pt KITTI-shaped) through SA+cluster path" if args.config == "kitti" else
                       "scenes/sec (65536-pt nuScenes-shaped, configs[4]) through SA+cluster path"),
            "value": round(world * B * steps / elapsed, 2),
            "unit": "scenes/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{'configs[1-2]' if args.config == 'kitti' else 'configs[4]'}: batch {B} x {cfg.n_points}-pt "
                                   f"{'KITTI' if args.config == 'kitti' else 'nuScenes'}-shaped scenes per GPU, "
                                   f"3-stage multi-radius SA backbone {'fp32' if args.dtype == 'f32' else 'bf16 (SPEC 14)'} + size-adaptive cluster layer + box head",
                       "scenes_per_gpu": B, "global_batch": world * B, "n_points": cfg.n_points,
                       "parallelism": f"batch-sharded x{world}, one all_gather of boxes",
                       "fps_overlap": not args.no_overlap, "fps_streams": args.fps_streams, "main_streams": args.main_streams, "opts": args.opt,
                       "mlp_geometry": tuned if tuned is not None else "heuristic"},
        }
        if log:
            tsteps = max(1, timed_steps)
            mlp_ms = per_kind.get("mlp", 0.0) / tsteps
            flops = work["mlp_flops"] * B                      # dense definition (SPEC.md §6)
            exec_flops, row_frac, per_flops = executed_flops(det, points, cfg)
            ach = exec_flops / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0
            n_mlp = sum(1 for k, _, _, _ in log if k == "mlp") // tsteps
            # the same launches without a sibling main stream (kernel durations are then not stretched
            # by the other batch's kernels): one main stream, sampling streams still overlapped
            ops.LAUNCH_LOG = []
            for _ in range(5):
                det(points, input_ready=True)
            torch.cuda.synchronize()
            iso_log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
            iso_ms = sum(e0.elapsed_time(e1) for k, _, e0, e1 in iso_log if k == "mlp") / 5
            iso = exec_flops / (iso_ms * 1e-3) / 1e12 if iso_ms > 0 else 0.0
            iso_name = {}
            for k, n, e0, e1 in iso_log:
                if k == "mlp":
                    iso_name[n] = iso_name.get(n, 0.0) + e0.elapsed_time(e1) / 5
            traffic = None
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_v12_pmc_traffic.json")
            if args.dtype == "f32" and args.config == "kitti" and os.path.exists(tpath):
                # HBM bytes of the same launches from the committed rocprofv3 --pmc passes (bench.py cannot
                # collect PMC counters itself): corrected FETCH_SIZE + WRITE_SIZE, per step like `achieved`
                tj = json.load(open(tpath))
                traffic = tj["fetch_bytes_per_step"] + tj["write_bytes_per_step"]
            res["roofline"] = {
                "kernel": f"{'mlp_chain_kernel' if args.dtype == 'f32' else 'mlp_bf16_kernel'} ({n_mlp} launches per step, summed)",
                "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK,
                "unit": "TFLOP/s", "frac": round(ach / PEAK, 4), "traffic": traffic,
                "traffic_note": "bytes per step from profiles/r01_v12_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                                "the step's MLP dispatches summed); ~1.1 GB per 2.8 ms = 0.4 TB/s: the kernel is MFMA-bound, not HBM-bound",
                "flop_per_step": exec_flops, "ms_per_step": round(mlp_ms, 3), "sampled_steps": tsteps,
                "note": "durations are HIP-event intervals on the launching stream inside the timed region, where "
                        "two main streams run consecutive batches side by side (kernels share the chip, so "
                        "their intervals stretch); achieved counts the flops the kernel EXECUTES: grouped rows that only repeat a "
                        "group's first neighbour (ball-query padding) are skipped exactly (a duplicate "
                        "row cannot change the max-pool), so executed < dense",
                "single_main_stream": {"ms_per_step": round(iso_ms, 3), "achieved": round(iso, 2),
                                       "frac": round(iso / PEAK, 4),
                                       "note": "same launches, consecutive batches NOT overlapped on a second main stream"},
                "dense_flop_per_step": flops, "executed_row_fraction": round(row_frac, 4),
                "dense_equivalent_tflops": round(flops / (mlp_ms * 1e-3) / 1e12, 2) if mlp_ms > 0 else 0.0}
            kern = []
            fps_ms = per_kind.get("fps", 0.0) / tsteps
            if fps_ms > 0:
                nested = getattr(det, "nested_fps_shortcut", True)
                serial = cfg.stages[0].npoint if nested else work["fps_steps"]
                kern.append({"kernel": "fps_sort_kernel + fps_cell_kernel (sampling streams; "
                                       + ("stage 1 only: stages 2-3 reuse its prefix, proven identical)" if nested else "3 stages)"),
                             "ms_per_step": round(fps_ms, 3),
                             "updates_per_s": round(work["fps_updates"] * B / (fps_ms * 1e-3) / 1e9, 2),
                             "unit": "G distance-updates/s (plain-scan equivalent; most are skipped exactly)",
                             "serial_steps": serial,
                             "us_per_serial_step": round(1e3 * fps_ms / serial, 3),
                             "bound": "serial latency (neither HBM nor MFMA)"})
            bq_ms = per_kind.get("ball_query", 0.0) / tsteps
            if bq_ms > 0:
                gbps = work["ball_query_bytes"] * B / (bq_ms * 1e-3) / 1e9
                kern.append({"kernel": "ball_query_kernel (4 launches per step)", "ms_per_step": round(bq_ms, 3),
                             "bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                             "frac": round(gbps / PEAK_HBM_GBPS, 5),
                             "pair_tests_per_s": round(work["pair_tests"] * B / (bq_ms * 1e-3) / 1e12, 3)})
            # the unfused group_points operator (not on the fused path; part of the drop-in surface):
            # an HBM-bound gather, timed here on the SA2 branch shape with its own HIP events
            gC, gN, gM, gS = 64, 4096, 1024, 32
            gfeat = torch.randn(B, gC, gN, device=dev)
            gidx = torch.randint(0, gN, (B, gM, gS), device=dev, dtype=torch.int32)
            ops.group_points(gfeat, gidx)
            torch.cuda.synchronize()
            ge0, ge1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ge0.record()
            for _ in range(10):
                ops.group_points(gfeat, gidx)
            ge1.record()
            torch.cuda.synchronize()
            g_ms = ge0.elapsed_time(ge1) / 10
            g_bytes = B * (gM * gS * 4 + gC * gM * gS * 4 + gC * gN * 4)     # idx + grouped out + source once
            kern.append({"kernel": "group_lds_kernel (unfused group_points, SA2 branch shape, outside the timed region)",
                         "ms_per_launch": round(g_ms, 4), "bound": "hbm",
                         "achieved": round(g_bytes / (g_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(g_bytes / (g_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                         "algorithmic_bytes": g_bytes})
            res["kernels"] = kern
            def fl(n):      # a merged dispatch is named "a+b+c"
                return sum(per_flops.get(x, 0) for x in n.split("+")) if "+" in n else per_flops.get(n, 0)
            res["mlp_launches"] = {n: {"ms": round(v / tsteps, 3), "executed_gflop": round(fl(n) / 1e9, 1),
                                       "tflops": round(fl(n) / (v / tsteps * 1e-3) / 1e12, 1),
                                       "ms_single_stream": round(iso_name.get(n, 0.0), 3),
                                       "tflops_single_stream": round(fl(n) / (iso_name[n] * 1e-3) / 1e12, 1) if iso_name.get(n) else None}
                                   for (k, n), v in sorted(per_name.items()) if k == "mlp"}
        if not args.no_cpu and world == 1:
            res["cpu_baseline"] = cpu_baseline(cfg, weights, args.cpu_scenes)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
