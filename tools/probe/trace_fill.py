"""Timeline of the first steps of the timed region from a kernel trace (trace_fill.sh): the timed region = the last run of 20
decode kernels; prints, relative to the first kernel after the gap in front of it, the first kernels per queue."""
import csv
import glob
import sys

out = sys.argv[1]
f = glob.glob(f"{out}/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dec = [i for i, r in enumerate(rows) if "decode_kernel" in r["Kernel_Name"]]
# the timed region's 20 decodes: the last 20 + whatever follows (parity pass): find the run of 20 decodes < 5 ms apart
ends = [int(rows[i]["End_Timestamp"]) for i in dec]
runs, cur = [], [0]
for k in range(1, len(dec)):
    if ends[k] - ends[k - 1] < 6_000_000:
        cur.append(k)
    else:
        runs.append(cur)
        cur = [k]
runs.append(cur)
run = [r for r in runs if len(r) == 20][-1]          # (the driver's setting: 20 timed steps; the warm-up / priming steps form other runs)
print("decode kernels per run:", [len(r) for r in runs])
last_dec_end = ends[run[-1]]
first_dec_end = ends[run[0]]
# start of the timed region: the first kernel after the longest idle gap in the 12 ms before the first decode of the run
i0 = dec[run[0]]
j = i0
best_gap, start_idx = 0, i0
prev_end = None
k = i0
while k > 0 and int(rows[k]["Start_Timestamp"]) > first_dec_end - 12_000_000:
    k -= 1
mx_end = max(int(r["End_Timestamp"]) for r in rows[:k + 1]) if k > 0 else 0
for q in range(k + 1, i0 + 1):
    s = int(rows[q]["Start_Timestamp"])
    if s - mx_end > best_gap:
        best_gap, start_idx = s - mx_end, q
    mx_end = max(mx_end, int(rows[q]["End_Timestamp"]))
t0 = int(rows[start_idx]["Start_Timestamp"])
print(f"timed region: starts after an idle gap of {best_gap / 1e3:.0f} us; {len(run)} steps in {(last_dec_end - t0) / 1e6:.3f} ms "
      f"= {(last_dec_end - t0) / 1e6 / len(run):.4f} ms/step; first decode at {(first_dec_end - t0) / 1e6:.3f} ms")
seen = {}
for r in rows[start_idx:]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    if s > 4200:
        break
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:38]
    key = (r["Queue_Id"], n)
    if key in seen:
        continue
    seen[key] = 1
    print(f"  t={s:8.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  q{r['Queue_Id']:>3}  {n}")
