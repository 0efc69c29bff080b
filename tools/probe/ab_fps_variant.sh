# pipelined step with the register-resident cell FPS (default) vs the record-streaming one (fps_variant=5: 47 VGPRs, 37 % of a CU's
# register file instead of 87 %) at several sampling-stream counts
run() { timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['step_ms']['p50'])"; }
run
run --opt fps_variant=5 --fps-streams 2
run --opt fps_variant=5 --fps-streams 3
run --opt fps_variant=5 --fps-streams 4
run --opt fps_variant=5 --fps-streams 6
run --fps-streams 3
