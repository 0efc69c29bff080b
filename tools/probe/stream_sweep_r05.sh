# stream-count sweep with step plans (the host no longer binds the bf16 step): bash tools/probe/stream_sweep_r05.sh OUTDIR
out=$1; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs --no-bf16-quality --steps 300 --warmup 24 "$@" > $out/$tag.json 2> $out/$tag.err; python - <<PY
import json
try:
    d = json.loads(open("$out/$tag.json").read().strip().splitlines()[-1])
    print(f"$tag: {d['value']:9.1f} scenes/s  {d['ms_per_step']:.4f} ms/step  p50 {d['step_ms']['p50']:.4f}  over2x {d['step_ms']['over_2x_p50']}", flush=True)
except Exception as e:
    print("$tag: FAILED", e, flush=True)
PY
}
for fs in 4 6 8; do for ms in 2 3; do run bf16_f${fs}_m${ms} --dtype bf16 --fps-streams $fs --main-streams $ms; done; done
run bf16_f6_m2_q12 --dtype bf16 --fps-streams 6 --main-streams 2 --queue-depth 11
for fs in 2 3 4; do run f32_f${fs}_m2 --fps-streams $fs --main-streams 2; done
run f32_f3_m3 --fps-streams 3 --main-streams 3
run nus_f6_m2 --config nuscenes --dtype bf16 --batches 2 --steps 80 --fps-streams 6
run nus_f8_m2 --config nuscenes --dtype bf16 --batches 2 --steps 80 --fps-streams 8
run nus_f6_m3 --config nuscenes --dtype bf16 --batches 2 --steps 80 --fps-streams 6 --main-streams 3
