# copies the summaries of gpurun_out/prof_r03 (tools/profile_r03.sh) into profiles/ under their round-3 names
src=gpurun_out/prof_r03; dst=profiles
cpif() { [ -s "$1" ] && cp "$1" "$2" && echo "$2"; }
cpif $src/bench.json $dst/r03_bench.json
cpif $src/bench_driver_setting.json $dst/r03_bench_driver_setting.json
cpif $src/bench_under_rocprof.json $dst/r03_bench_under_rocprof.json
cpif $src/geometry.json $dst/r03_geometry.json
cpif $src/roofline_from_profiles.txt $dst/r03_roofline_from_profiles.txt
cpif $src/serial_kernel_stats.csv $dst/r03_serial_kernel_stats.csv
cpif $src/serial_kernel_trace.csv $dst/r03_serial_kernel_trace.csv
cpif $src/timed_kernel_stats.csv $dst/r03_timed_kernel_stats.csv
if ls $src/pmc_fetch/*/*_counter_collection.csv >/dev/null 2>&1; then
  tmp=$(mktemp -d); for n in fetch write sq; do ln -s $(pwd)/$src/pmc_$n $tmp/$n; done
  python3 tools/pmc_summary.py $tmp > $dst/r03_pmc_summary.txt && echo $dst/r03_pmc_summary.txt
  python3 tools/pmc_traffic.py $tmp $dst/r03_pmc_traffic.json r03 > /dev/null && echo $dst/r03_pmc_traffic.json
  rm -rf $tmp
fi
cpif $src/bf16_bench.json $dst/r03_bf16_bench.json
cpif $src/bf16_geometry.json $dst/r03_bf16_geometry.json
cpif $src/bf16_roofline_from_profiles.txt $dst/r03_bf16_roofline_from_profiles.txt
cpif $src/bf16_serial_kernel_stats.csv $dst/r03_bf16_serial_kernel_stats.csv
cpif $src/bf16_serial_kernel_trace.csv $dst/r03_bf16_serial_kernel_trace.csv
if ls $src/pmc_bf16sq/*/*_counter_collection.csv >/dev/null 2>&1; then
  tmp=$(mktemp -d); ln -s $(pwd)/$src/pmc_bf16sq $tmp/sq
  python3 tools/pmc_summary.py $tmp > $dst/r03_bf16_pmc_summary.txt && echo $dst/r03_bf16_pmc_summary.txt
  rm -rf $tmp
fi
cpif $src/nuscenes_bf16_bench.json $dst/r03_nuscenes_bf16_bench.json
cpif $src/nus_geometry.json $dst/r03_nuscenes_bf16_geometry.json
cpif $src/nuscenes_bf16_roofline_from_profiles.txt $dst/r03_nuscenes_bf16_roofline_from_profiles.txt
cpif $src/nus_serial_kernel_stats.csv $dst/r03_nuscenes_bf16_serial_kernel_stats.csv
cpif $src/nus_serial_kernel_trace.csv $dst/r03_nuscenes_bf16_serial_kernel_trace.csv
