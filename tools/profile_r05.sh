# Round-5 evidence for profiles/ (run on the GPU box through gpurun; ~8 minutes).  usage: bash tools/profile_r05.sh [part...]
# parts: f32 pmc bf16 nus (default: all).  tools/collect_r05.sh copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r05; mkdir -p $out
parts="${*:-f32 pmc bf16 nus}"
trim() {  # trim <dir with rocprof output> <dest csv>: name / start / end / grid of every kernel, time-ordered
python3 - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "Scratch_Size"]
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w", newline="") as g:
    w = csv.DictWriter(g, keep); w.writeheader()
    for r in rows:
        n = r["Kernel_Name"]
        if "at::native" in n: n = n.split("<")[0][:60]
        w.writerow({**{k: r[k] for k in keep}, "Kernel_Name": n, "Start_Timestamp": int(r["Start_Timestamp"]) - t0, "End_Timestamp": int(r["End_Timestamp"]) - t0})
PY
}
pb() { python3 -c "import json,sys; print(json.load(open(sys.argv[1]))['config']['parity_batch'])" "$1"; }
serial() {  # serial <name> <bench json whose parity batch is traced> <bench args...>: one stream, nothing overlapped, ONE batch
  n=$1; src=$2; shift 2
  rm -rf $out/$n
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$n -- python3 bench.py --no-cpu --no-dense-leg --no-legs --no-launch-timing --no-bf16-quality --main-streams 1 --no-overlap --batches 1 --first-batch $(pb $src) --steps 20 --warmup 2 "$@" > $out/$n.log 2>&1 || { tail -3 $out/$n.log; return 1; }
  trim $out/$n $out/${n}_kernel_trace.csv
  cp $(ls $out/$n/*/*kernel_stats.csv | head -1) $out/${n}_kernel_stats.csv
}
pmc() {  # pmc <name> <counters...> -- <bench args...>: counters in their own pass (kernel-trace only, as gpurun requires)
  n=$1; shift; ctr=""; while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  rm -rf $out/pmc_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_$n -- python3 bench.py --no-cpu --no-dense-leg --no-legs --no-launch-timing --no-bf16-quality --main-streams 1 --batches 1 --steps 3 --warmup 1 "$@" > $out/pmc_$n.log 2>&1 || { echo "pass $n failed"; tail -5 $out/pmc_$n.log; return 1; }
}
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
for part in $parts; do case $part in
f32)
  # the default bench line (rotating batches, legs, CPU leg), saving the geometry it used; the driver's setting; rocprof of the timed configuration
  python3 bench.py --save-geometry $out/geometry.json > $out/bench.log 2>&1 || { tail -3 $out/bench.log; exit 1; }
  grep -E '^\{' $out/bench.log > $out/bench.json
  python3 bench.py --steps 20 --warmup 5 --geometry-file $out/geometry.json 2>/dev/null | grep -E '^\{' > $out/bench_driver_setting.json
  rm -rf $out/timed
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/timed -- python3 bench.py --no-cpu --no-dense-leg --no-legs --geometry-file $out/geometry.json > $out/timed.log 2>&1 || { tail -3 $out/timed.log; exit 1; }
  grep -E '^\{' $out/timed.log > $out/bench_under_rocprof.json
  cp $(ls $out/timed/*/*kernel_stats.csv | head -1) $out/timed_kernel_stats.csv
  serial serial $out/bench.json --geometry-file $out/geometry.json || exit 1
  python3 tools/roofline_from_profiles.py $out/serial_kernel_trace.csv $out/bench.json 20 | tee $out/roofline_from_profiles.txt
  ;;
pmc)
  pmc fetch FETCH_SIZE -- --geometry-file $out/geometry.json && pmc write WRITE_SIZE -- --geometry-file $out/geometry.json &&
  pmc sq $SQ SQ_INSTS_VALU_MFMA_MOPS_F32 -- --geometry-file $out/geometry.json
  ;;
bf16)
  python3 bench.py --dtype bf16 --no-cpu --no-legs --save-geometry $out/bf16_geometry.json > $out/bf16_bench.log 2>&1 || { tail -3 $out/bf16_bench.log; exit 1; }
  grep -E '^\{' $out/bf16_bench.log > $out/bf16_bench.json
  serial bf16_serial $out/bf16_bench.json --dtype bf16 --geometry-file $out/bf16_geometry.json || exit 1
  python3 tools/roofline_from_profiles.py $out/bf16_serial_kernel_trace.csv $out/bf16_bench.json 20 | tee $out/bf16_roofline_from_profiles.txt
  pmc bf16fetch FETCH_SIZE -- --dtype bf16 --geometry-file $out/bf16_geometry.json && pmc bf16write WRITE_SIZE -- --dtype bf16 --geometry-file $out/bf16_geometry.json
  pmc bf16sq $SQ -- --dtype bf16 --no-overlap --geometry-file $out/bf16_geometry.json
  ;;
nus)
  python3 bench.py --config nuscenes --dtype bf16 --batch 32 --batches 2 --no-cpu --no-legs --steps 80 --warmup 8 --save-geometry $out/nus_geometry.json > $out/nus_bench.log 2>&1 || { tail -3 $out/nus_bench.log; exit 1; }
  grep -E '^\{' $out/nus_bench.log > $out/nuscenes_bf16_bench.json
  rm -rf $out/nus_serial
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/nus_serial -- python3 bench.py --config nuscenes --dtype bf16 --batch 32 --no-cpu --no-dense-leg --no-legs --no-launch-timing --no-bf16-quality --main-streams 1 --no-overlap --batches 1 --first-batch $(pb $out/nuscenes_bf16_bench.json) --steps 4 --warmup 2 --geometry-file $out/nus_geometry.json > $out/nus_serial.log 2>&1 || { tail -3 $out/nus_serial.log; exit 1; }
  trim $out/nus_serial $out/nus_serial_kernel_trace.csv
  cp $(ls $out/nus_serial/*/*kernel_stats.csv | head -1) $out/nus_serial_kernel_stats.csv
  python3 tools/roofline_from_profiles.py $out/nus_serial_kernel_trace.csv $out/nuscenes_bf16_bench.json 4 | tee $out/nuscenes_bf16_roofline_from_profiles.txt
  pmc nusfetch FETCH_SIZE -- --config nuscenes --dtype bf16 --geometry-file $out/nus_geometry.json && pmc nuswrite WRITE_SIZE -- --config nuscenes --dtype bf16 --geometry-file $out/nus_geometry.json
  ;;
esac; done
ls -la $out/*.csv $out/*.json $out/*.txt 2>/dev/null
