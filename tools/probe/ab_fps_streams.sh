run() { timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['step_ms']['p50'], d['config']['queue_depth'])"; }
for i in 1 2; do
run --fps-streams 2
run --fps-streams 3
run --fps-streams 4
run --fps-streams 3 --queue-depth 8
run --fps-streams 3 --main-streams 3
done
run --fps-streams 3 --steps 20 --warmup 5
run --fps-streams 2 --steps 20 --warmup 5
run --fps-streams 4 --steps 20 --warmup 5
