"""Per-queue view of a kernel trace of a pipelined bench run: busy time per step, gaps between consecutive kernels of a queue (count and
sum by size class), over the middle of the longest run of steps.  usage: python tools/probe/trace_queues.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dec = [int(r["End_Timestamp"]) for r in rows if "decode_kernel" in r["Kernel_Name"]]
best, cur = (0, 0), 0
for i in range(1, len(dec) + 1):
    if i == len(dec) or dec[i] - dec[i - 1] > 8_000_000:
        if i - cur > best[1] - best[0]:
            best = (cur, i)
        cur = i
n = best[1] - best[0]
a, b = best[0] + n // 5, best[1] - n // 5
t0, t1, steps = dec[a], dec[b], b - a
rows = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
print(f"{steps} steps in {(t1 - t0) / 1e6:.2f} ms = {(t1 - t0) / 1e6 / steps:.4f} ms/step")
perq = defaultdict(list)
for r in rows:
    perq[int(r["Queue_Id"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, iv in sorted(perq.items()):
    busy = sum(e - s for s, e, _ in iv) / 1e6
    gaps = [(iv[i + 1][0] - iv[i][1]) / 1e3 for i in range(len(iv) - 1)]
    cls = [(0, 2), (2, 10), (10, 30), (30, 100), (100, 1e9)]
    txt = "  ".join(f"{lo:g}-{hi:g}us: {sum(1 for g in gaps if lo <= g < hi) / steps:.1f}/step {sum(g for g in gaps if lo <= g < hi) / steps:.0f}us" for lo, hi in cls[:-1])
    big = sum(g for g in gaps if g >= 100) / steps
    names = {}
    for s, e, nme in iv:
        names[nme.split("(")[0][-28:]] = names.get(nme.split("(")[0][-28:], 0) + 1
    top = sorted(names.items(), key=lambda t: -t[1])[:2]
    print(f"queue {q:2d} (class {q % 4}): {len(iv) / steps:5.1f} kernels/step, busy {busy / steps * 1e3:6.0f} us/step | gaps {txt}  >=100us: {big:.0f}us | {top}")
