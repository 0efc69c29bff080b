# copies the summaries of gpurun_out/prof_r05 (tools/profile_r05.sh) into profiles/ under their round-5 names
src=gpurun_out/prof_r05; dst=profiles
cpif() { [ -s "$1" ] && cp "$1" "$2" && echo "$2"; }
cpif $src/bench.json $dst/r05_bench.json
cpif $src/bench_driver_setting.json $dst/r05_bench_driver_setting.json
cpif $src/bench_under_rocprof.json $dst/r05_bench_under_rocprof.json
cpif $src/geometry.json $dst/r05_geometry.json
cpif $src/roofline_from_profiles.txt $dst/r05_roofline_from_profiles.txt
cpif $src/serial_kernel_stats.csv $dst/r05_serial_kernel_stats.csv
cpif $src/serial_kernel_trace.csv $dst/r05_serial_kernel_trace.csv
cpif $src/timed_kernel_stats.csv $dst/r05_timed_kernel_stats.csv
traffic() {  # traffic <fetch pass> <write pass> <sq pass or ""> <prefix>
  if ls $src/pmc_$1/*/*_counter_collection.csv >/dev/null 2>&1; then
    tmp=$(mktemp -d); ln -s $(pwd)/$src/pmc_$1 $tmp/fetch; ln -s $(pwd)/$src/pmc_$2 $tmp/write; [ -n "$3" ] && ln -s $(pwd)/$src/pmc_$3 $tmp/sq
    [ -n "$3" ] && python3 tools/pmc_summary.py $tmp > $dst/$4pmc_summary.txt && echo $dst/$4pmc_summary.txt
    python3 tools/pmc_traffic.py $tmp $dst/$4pmc_traffic.json r05 > /dev/null && echo $dst/$4pmc_traffic.json
    rm -rf $tmp
  fi
}
traffic fetch write sq r05_
cpif $src/bf16_bench.json $dst/r05_bf16_bench.json
cpif $src/bf16_geometry.json $dst/r05_bf16_geometry.json
cpif $src/bf16_roofline_from_profiles.txt $dst/r05_bf16_roofline_from_profiles.txt
cpif $src/bf16_serial_kernel_stats.csv $dst/r05_bf16_serial_kernel_stats.csv
cpif $src/bf16_serial_kernel_trace.csv $dst/r05_bf16_serial_kernel_trace.csv
traffic bf16fetch bf16write bf16sq r05_bf16_
cpif $src/nuscenes_bf16_bench.json $dst/r05_nuscenes_bf16_bench.json
cpif $src/nus_geometry.json $dst/r05_nuscenes_bf16_geometry.json
cpif $src/nuscenes_bf16_roofline_from_profiles.txt $dst/r05_nuscenes_bf16_roofline_from_profiles.txt
cpif $src/nus_serial_kernel_stats.csv $dst/r05_nuscenes_bf16_serial_kernel_stats.csv
cpif $src/nus_serial_kernel_trace.csv $dst/r05_nuscenes_bf16_serial_kernel_trace.csv
traffic nusfetch nuswrite "" r05_nuscenes_bf16_
