# nuScenes-shaped bf16 (configs[4]): scenes per batch x sampling streams
run() { timeout -k 10 280 python bench.py --config nuscenes --dtype bf16 --no-cpu --no-dense-leg --no-launch-timing --steps 40 --warmup 6 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['config'].get('queue_depth'))"; }
run --batch 8 --fps-streams 8
run --batch 8 --fps-streams 12
run --batch 8 --fps-streams 16
run --batch 16 --fps-streams 6
run --batch 16 --fps-streams 8
run --batch 16 --fps-streams 10
run --batch 32 --fps-streams 3
run --batch 32 --fps-streams 4
run --batch 32 --fps-streams 5
run --batch 32 --fps-streams 6
