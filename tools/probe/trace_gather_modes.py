"""Compare two kernel traces (trace_gather_modes.sh): per kernel name the mean duration, and per queue the busy time and the
gaps between consecutive kernels, over the steady-state part of the timed region (found from the decode kernels: the longest
run of steps less than 5 ms apart, its middle 60 %)."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for m in ("none", "full"):
    f = glob.glob(f"{out}/{m}/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dec = [int(r["End_Timestamp"]) for r in rows if "decode_kernel" in r["Kernel_Name"]]
    best, cur = (0, 0), 0
    for i in range(1, len(dec) + 1):
        if i == len(dec) or dec[i] - dec[i - 1] > 5_000_000:
            if i - cur > best[1] - best[0]:
                best = (cur, i)
            cur = i
    n = best[1] - best[0]
    a, b = best[0] + n // 5, best[1] - n // 5
    t0, t1 = dec[a], dec[b]
    steps = b - a
    rows = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
    span = (t1 - t0) / 1e6
    dur = defaultdict(list)
    perq = defaultdict(list)
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        dur[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]].append(d)
        perq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    res[m] = (span, dur, perq, steps)
    print(m, f"{steps} steps in {span:.2f} ms = {span / steps:.4f} ms/step; kernels {len(rows)}, queues {len(perq)}")
    for q, iv in sorted(perq.items()):
        busy = sum(e - s for s, e in iv) / 1e6
        gaps = [(iv[i + 1][0] - iv[i][1]) / 1e3 for i in range(len(iv) - 1)]
        small = [g for g in gaps if 0 <= g < 50]
        print(f"   queue {q}: {len(iv)} kernels, busy {busy / steps:.3f} ms/step, gaps < 50 us: {sum(small) / steps:.1f} us/step ({len(small)}), "
              f"median {sorted(gaps)[len(gaps)//2]:.2f} us")
print(f"{'kernel':46s} n/step(none) us(none) n/step(full) us(full)  d us/step")
names = sorted(res["none"][1], key=lambda k: -sum(res["none"][1][k]))
tot = 0.0
for k in names:
    a, b = res["none"][1][k], res["full"][1].get(k, [])
    sa, sb = res["none"][3], res["full"][3]
    d = (sum(b) / sb if b else 0.0) - sum(a) / sa
    tot += d
    if abs(d) > 1.0 or names.index(k) < 12:
        print(f"{k:46s} {len(a)/sa:8.2f} {sum(a)/len(a):10.1f} {len(b)/sb:8.2f} {(sum(b)/len(b)) if b else 0:10.1f}   {d:8.1f}")
print("sum of kernel time per step, full - none:", round(tot, 1), "us")
