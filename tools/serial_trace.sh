# in-situ kernel durations of one serial step (one stream, nothing overlapped) under rocprofv3: bash tools/serial_trace.sh <tag> [bench args]
tag=${1:-st}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/serial -- python3 bench.py --no-cpu --no-dense-leg --no-launch-timing --main-streams 1 --no-overlap --steps 20 --warmup 2 --geometry-file profiles/r02_geometry.json "$@" > $out/serial.log 2>&1 || { tail -3 $out/serial.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/serial/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    d[(n[:44], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 10: print(f"{n:44s} grid {g:>8s} x{len(v):4d}  median {sorted(v)[len(v)//2]:8.1f} us  sum/20 {sum(v)/20:8.1f}")
PY
