# kernel trace of one python tool run: bash tools/trace_one.sh <name> <script and args>
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/$name; mkdir -p gpurun_out/$name
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$name -- python3 "$@" > gpurun_out/$name/run.log 2>&1 || { tail -5 gpurun_out/$name/run.log; exit 1; }
python3 - gpurun_out/$name <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("mlp_", "rowscan", "fillBuffer", "FillBuffer", "memset", "grid_", "ball_query", "fps_"))]
tail = rows[-14:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    n = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid {r['Grid_Size_X']:>8s}  {n[:50]}")
PY
