# Round-2 evidence for profiles/ (run on the GPU box through gpurun; ~3 minutes).  usage: bash tools/profile_r02.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
# 1. the default bench line (autotuned), saving the geometry it used
python3 bench.py --save-geometry $out/geometry.json > $out/bench.log 2>&1 || { tail -3 $out/bench.log; exit 1; }
grep -E '^\{' $out/bench.log > $out/bench.json
# 2. the same command under rocprofv3 (kernel stats of the timed configuration; geometry from the file: no autotune launches)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/timed -- python3 bench.py --no-cpu --no-dense-leg --geometry-file $out/geometry.json > $out/timed.log 2>&1 || { tail -3 $out/timed.log; exit 1; }
grep -E '^\{' $out/timed.log > $out/bench_under_rocprof.json
# 3. serial run (one stream, nothing overlapped): kernel intervals = kernel durations
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -- python3 bench.py --no-cpu --no-dense-leg --no-launch-timing --main-streams 1 --no-overlap --steps 20 --warmup 2 --geometry-file $out/geometry.json > $out/serial.log 2>&1 || { tail -3 $out/serial.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/serial/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "Scratch_Size"]
t0 = int(rows[0]["Start_Timestamp"])
with open(out + "/serial_kernel_trace.csv", "w", newline="") as g:
    w = csv.DictWriter(g, keep); w.writeheader()
    for r in rows:
        n = r["Kernel_Name"]
        if "at::native" in n: n = n.split("<")[0][:60]
        w.writerow({**{k: r[k] for k in keep}, "Kernel_Name": n, "Start_Timestamp": int(r["Start_Timestamp"]) - t0, "End_Timestamp": int(r["End_Timestamp"]) - t0})
PY
cp $(ls $out/timed/*/*kernel_stats.csv | head -1) $out/timed_kernel_stats.csv
cp $(ls $out/serial/*/*kernel_stats.csv | head -1) $out/serial_kernel_stats.csv
python3 tools/roofline_from_profiles.py $out/serial_kernel_trace.csv $out/bench.json 20 | tee $out/roofline_from_profiles.txt
ls -la $out/*.csv $out/*.json
