# Round-3 evidence for profiles/ (run on the GPU box through gpurun; ~6 minutes).  usage: bash tools/profile_r03.sh [part...]
# parts: f32 pmc bf16 nus (default: all)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r03; mkdir -p $out
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
serial() {  # serial <name> <bench args...>: one stream, nothing overlapped -> kernel intervals = kernel durations
  n=$1; shift
  rm -rf $out/$n
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$n -- python3 bench.py --no-cpu --no-dense-leg --no-launch-timing --main-streams 1 --no-overlap --steps 20 --warmup 2 "$@" > $out/$n.log 2>&1 || { tail -3 $out/$n.log; return 1; }
  trim $out/$n $out/${n}_kernel_trace.csv
  cp $(ls $out/$n/*/*kernel_stats.csv | head -1) $out/${n}_kernel_stats.csv
}
for part in $parts; do case $part in
f32)
  # the default bench line (autotuned, with the CPU leg), saving the geometry it used; the driver's setting; rocprof of the timed configuration
  python3 bench.py --save-geometry $out/geometry.json > $out/bench.log 2>&1 || { tail -3 $out/bench.log; exit 1; }
  grep -E '^\{' $out/bench.log > $out/bench.json
  python3 bench.py --steps 20 --warmup 5 --no-cpu --geometry-file $out/geometry.json 2>/dev/null | grep -E '^\{' > $out/bench_driver_setting.json
  rm -rf $out/timed
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/timed -- python3 bench.py --no-cpu --no-dense-leg --geometry-file $out/geometry.json > $out/timed.log 2>&1 || { tail -3 $out/timed.log; exit 1; }
  grep -E '^\{' $out/timed.log > $out/bench_under_rocprof.json
  cp $(ls $out/timed/*/*kernel_stats.csv | head -1) $out/timed_kernel_stats.csv
  serial serial --geometry-file $out/geometry.json || exit 1
  python3 tools/roofline_from_profiles.py $out/serial_kernel_trace.csv $out/bench.json 20 | tee $out/roofline_from_profiles.txt
  ;;
pmc)
  # counters in their own passes (kernel-trace only, as gpurun requires)
  run() { n=$1; shift
    rm -rf $out/pmc_$n
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/pmc_$n -- python3 bench.py --no-cpu --no-dense-leg --geometry-file $out/geometry.json --no-launch-timing --main-streams 1 --steps 3 --warmup 1 > $out/pmc_$n.log 2>&1 || { echo "pass $n failed"; tail -5 $out/pmc_$n.log; return 1; }
  }
  run fetch FETCH_SIZE && run write WRITE_SIZE &&
  run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
  ;;
bf16)
  python3 bench.py --dtype bf16 --no-cpu --save-geometry $out/bf16_geometry.json > $out/bf16_bench.log 2>&1 || { tail -3 $out/bf16_bench.log; exit 1; }
  grep -E '^\{' $out/bf16_bench.log > $out/bf16_bench.json
  serial bf16_serial --dtype bf16 --geometry-file $out/bf16_geometry.json || exit 1
  python3 tools/roofline_from_profiles.py $out/bf16_serial_kernel_trace.csv $out/bf16_bench.json 20 | tee $out/bf16_roofline_from_profiles.txt
  rm -rf $out/pmc_bf16sq
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_bf16sq -- python3 bench.py --dtype bf16 --no-cpu --geometry-file $out/bf16_geometry.json --no-launch-timing --main-streams 1 --no-overlap --steps 3 --warmup 1 > $out/pmc_bf16sq.log 2>&1 || tail -3 $out/pmc_bf16sq.log
  ;;
nus)
  python3 bench.py --config nuscenes --dtype bf16 --batch 32 --no-cpu --steps 40 --warmup 6 --save-geometry $out/nus_geometry.json > $out/nus_bench.log 2>&1 || { tail -3 $out/nus_bench.log; exit 1; }
  grep -E '^\{' $out/nus_bench.log > $out/nuscenes_bf16_bench.json
  serial nus_serial --config nuscenes --dtype bf16 --batch 32 --steps 4 --geometry-file $out/nus_geometry.json || exit 1
  python3 tools/roofline_from_profiles.py $out/nus_serial_kernel_trace.csv $out/nuscenes_bf16_bench.json 4 | tee $out/nuscenes_bf16_roofline_from_profiles.txt
  ;;
esac; done
ls -la $out/*.csv $out/*.json $out/*.txt 2>/dev/null
