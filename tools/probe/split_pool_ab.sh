# Split pooling on / off (SAD_NO_SPLIT_POOL=1), alternating runs of the pipelined bf16 step at one saved geometry inside ONE gpurun call:
#   bash tools/probe/split_pool_ab.sh OUTDIR ROUNDS "bench args"      (e.g. "--dtype bf16" or "--config nuscenes --dtype bf16 --steps 80")
out=$1; rounds=$2; bargs=$3
mkdir -p $out
geom=$out/geometry.json
if [ ! -f $geom ]; then
  timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs --no-bf16-quality --steps 50 --warmup 5 $bargs --save-geometry $geom > $out/tune.json 2> $out/tune.err || { echo "tune run failed"; tail -5 $out/tune.err; exit 1; }
fi
for r in $(seq 1 $rounds); do
  for mode in split f32pool; do
    E=""; [ $mode = f32pool ] && E=1
    SAD_NO_SPLIT_POOL=$E timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs --no-bf16-quality --steps ${STEPS:-400} --warmup 10 $bargs --geometry-file $geom > $out/${mode}_$r.json 2> $out/${mode}_$r.err
    python - <<PY
import json
try:
    d = json.loads(open("$out/${mode}_$r.json").read().strip().splitlines()[-1])
    print(f"$mode round $r: {d['value']:9.1f} scenes/s  {d['ms_per_step']:.4f} ms/step  p50 {d['step_ms']['p50']:.4f}  plans {d['config'].get('step_plans')}", flush=True)
except Exception as e:
    print("$mode round $r: FAILED", e, flush=True)
PY
  done
done
