# One sad_set_option knob, alternating values, pipelined runs at one saved geometry inside ONE gpurun call:
#   bash tools/probe/opt_ab.sh OUTDIR ROUNDS "bench args" KEY v1 v2 ...      e.g. ... "--dtype bf16" mlp_rows_form 0 1
out=$1; rounds=$2; bargs=$3; key=$4; shift 4
mkdir -p $out
geom=$out/geometry.json
if [ ! -f $geom ]; then
  timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs --no-bf16-quality --steps 50 --warmup 5 $bargs --save-geometry $geom > $out/tune.json 2> $out/tune.err || { echo "tune run failed"; tail -5 $out/tune.err; exit 1; }
fi
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs --no-bf16-quality --steps ${STEPS:-400} --warmup 10 $bargs --geometry-file $geom --opt $key=$v > $out/${key}${v}_$r.json 2> $out/${key}${v}_$r.err
    python - <<PY
import json
try:
    d = json.loads(open("$out/${key}${v}_$r.json").read().strip().splitlines()[-1])
    print(f"$key=$v round $r: {d['value']:9.1f} scenes/s  {d['ms_per_step']:.4f} ms/step  p50 {d['step_ms']['p50']:.4f}  parity {d.get('parity_check', {}).get('ok')}", flush=True)
except Exception as e:
    print("$key=$v round $r: FAILED", e, flush=True)
PY
  done
done
