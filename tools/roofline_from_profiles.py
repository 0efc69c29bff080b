#!/usr/bin/env python3
"""Recompute the roofline fraction of the MLP dispatches from COMMITTED rocprofv3 evidence alone.

    python tools/roofline_from_profiles.py profiles/r02_serial_kernel_trace.csv profiles/r02_bench.json

Inputs (both produced by tools/profile_r02.sh and committed under profiles/):
  * a kernel trace (rocprofv3 --kernel-trace, trimmed to name / start / end / grid by the script) of
    `bench.py --main-streams 1 --no-overlap --geometry-file ...`: ONE stream, nothing overlapped, no autotune
    launches, so every kernel interval is a kernel duration and the trace holds only steady-state steps;
  * the JSON line of a normal bench run of the same build: `mlp_launch_order` (names of the MLP dispatches of
    a step, in launch order) and `mlp_launches[name].executed_gflop` (flops the kernels execute — ball-query
    padding rows are skipped exactly; the batch is the same seeded synthetic batch in both runs).
Per step the MLP kernels are grouped into dispatches in launch order: a row-packing scan
(rowscan_sums + rowscan_write) that is DIRECTLY followed by an MLP kernel was launched by that dispatch and opens
it (the cluster layer); a scan followed by other kernels first (the SA stages: it runs behind the ball query, on
the sampling stream in the timed configuration) is not part of any MLP dispatch; consecutive mlp_layer_kernel
launches (one per layer) belong together, every other mlp_* kernel is its own dispatch.  Prints per-dispatch duration and
TFLOP/s and the total fraction of the dense MFMA peak of the bench line's dtype (f32 157.3, bf16 2 500 TFLOP/s)."""
import csv
import json
import sys

PEAK_F32, PEAK_BF16 = 157.3, 2500.0      # dense MFMA peaks, TFLOP/s (/opt/skills/guides/MI355X_MICROARCH.md)


def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].strip()


def main():
    trace, bench = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    allk = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
    ks = []
    for i, (n, d) in enumerate(allk):
        if n.startswith(("bf16_rows_kernel", "bf16_rows2_kernel")):       # (the plain-row layer of the bf16 mode, both forms: an MLP kernel like the mlp_* ones)
            n = "mlp_" + n
        if n == "mlp_pack_kernel" or not n.startswith(("mlp_", "rowscan_")):
            continue
        if n == "rowscan_sums_kernel":      # part of an MLP dispatch only if sums, write, MLP kernel follow each other directly
            own = i + 2 < len(allk) and allk[i + 1][0] == "rowscan_write_kernel" and allk[i + 2][0].startswith("mlp_")
            if not own:
                continue
        if n == "rowscan_write_kernel" and not (ks and ks[-1][0] == "rowscan_sums_kernel" and i > 0 and allk[i - 1][0] == "rowscan_sums_kernel"):
            continue
        ks.append((n, d))
    groups, cur = [], None          # a group = [names..., total us]
    for n, d in ks:
        if n == "rowscan_sums_kernel":
            cur = {"kernels": [n], "us": d, "open": True, "layer": False}
            groups.append(cur)
        elif n == "rowscan_write_kernel":
            cur["kernels"].append(n); cur["us"] += d
        elif n == "mlp_layer_kernel":
            if cur is not None and cur["open"] and (cur["layer"] or cur["kernels"][-1].startswith("rowscan")):
                cur["kernels"].append(n); cur["us"] += d; cur["layer"] = True
            else:
                cur = {"kernels": [n], "us": d, "open": True, "layer": True}
                groups.append(cur)
        else:
            if cur is not None and cur["open"] and not cur["layer"] and cur["kernels"][-1].startswith("rowscan"):
                cur["kernels"].append(n); cur["us"] += d; cur["open"] = False
            else:
                cur = {"kernels": [n], "us": d, "open": False, "layer": False}
                groups.append(cur)
    j = None
    for line in open(bench):
        if line.startswith("{"):
            j = json.loads(line)
    PEAK = PEAK_BF16 if j.get("dtype") == "bf16" else PEAK_F32
    order = j["mlp_launch_order"]
    flops = {n: v["executed_gflop"] * 1e9 for n, v in j["mlp_launches"].items()}
    per = len(order)
    nsteps = steps or len(groups) // per
    use = groups[len(groups) - nsteps * per:]
    assert len(use) == nsteps * per and nsteps >= 1, (len(groups), per)
    tot_us, tot_fl = 0.0, 0.0
    print(f"{nsteps} steady-state steps x {per} MLP dispatches from {trace}")
    print(f"{'dispatch':34s} {'us':>9s} {'GFLOP':>8s} {'TFLOP/s':>8s} {'frac':>6s}  kernels")
    for i, name in enumerate(order):
        g = [use[s * per + i] for s in range(nsteps)]
        us = sum(x["us"] for x in g) / nsteps
        fl = flops[name]
        tot_us += us; tot_fl += fl
        print(f"{name:34s} {us:9.1f} {fl / 1e9:8.2f} {fl / us / 1e6:8.1f} {fl / us / 1e6 / PEAK:6.3f}  {' + '.join(g[0]['kernels'])}")
    print(f"{'all MLP dispatches of a step':34s} {tot_us:9.1f} {tot_fl / 1e9:8.2f} {tot_fl / tot_us / 1e6:8.1f} {tot_fl / tot_us / 1e6 / PEAK:6.3f}")
    print(f"roofline.frac recomputed from the trace: {tot_fl / tot_us / 1e6 / PEAK:.4f}   (bench line: {j['roofline']['frac']})")


if __name__ == "__main__":
    main()
