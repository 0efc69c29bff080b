# Workgroups per CU of the persistent bf16 chain kernel (mlp_dyn_slots), alternating runs at one saved geometry:
#   bash tools/probe/bf16_slots_ab.sh OUTDIR ROUNDS "bench args" slots...     (0 = the occupancy the runtime reports)
out=$1; rounds=$2; bargs=$3; shift 3
mkdir -p $out
geom=$out/geometry.json
if [ ! -f $geom ]; then
  timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs --no-bf16-quality --steps 50 --warmup 5 $bargs --save-geometry $geom > $out/tune.json 2> $out/tune.err || { echo "tune run failed"; tail -5 $out/tune.err; exit 1; }
fi
for r in $(seq 1 $rounds); do
  for k in "$@"; do
    timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs --no-bf16-quality --steps ${STEPS:-400} --warmup 10 $bargs --geometry-file $geom --opt mlp_dyn_slots=$k > $out/s${k}_$r.json 2> $out/s${k}_$r.err
    python - <<PY
import json
try:
    d = json.loads(open("$out/s${k}_$r.json").read().strip().splitlines()[-1])
    print(f"slots $k round $r: {d['value']:9.1f} scenes/s  {d['ms_per_step']:.4f} ms/step  p50 {d['step_ms']['p50']:.4f}", flush=True)
except Exception as e:
    print("slots $k round $r: FAILED", e, flush=True)
PY
  done
done
