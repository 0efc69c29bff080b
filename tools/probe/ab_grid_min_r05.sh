# SA3 ball query (1 024 points) through the grid instead of the scan: pipelined f32 and bf16 steps, alternating, one saved geometry per dtype
out=gpurun_out/r5r; mkdir -p $out
for dt in f32 bf16; do
  python bench.py --dtype $dt --no-cpu --no-dense-leg --no-launch-timing --no-legs --no-bf16-quality --steps 50 --warmup 5 --save-geometry $out/geom_$dt.json > /dev/null 2>&1
  for r in 1 2 3; do for thr in 2048 1024; do
    python tools/probe/bench_with_grid_min.py $thr -- --dtype $dt --no-cpu --no-dense-leg --no-launch-timing --no-legs --no-bf16-quality --steps 300 --warmup 10 --geometry-file $out/geom_$dt.json 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt grid_min $thr round $r:', d['value'], d['ms_per_step'], d['step_ms']['p50'], d['config']['step_plans']['refused'])"
  done; done
done
