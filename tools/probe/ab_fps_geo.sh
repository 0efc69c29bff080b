run() { timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['step_ms']['p50'])"; }
for i in 1 2; do
run
run --opt fps_threads=832
run --dtype bf16
run --dtype bf16 --opt fps_threads=832
done
