# bf16 KITTI-shaped standalone runs, N rounds: bash tools/probe/bf16_default_ab.sh [rounds]
for r in $(seq 1 ${1:-3}); do
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu --no-legs --no-dense-leg --no-bf16-quality --steps 400 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms']['p50'], d['config']['mlp_geometry'], {n: v['ms'] for n, v in d['mlp_launches'].items() if n in ('cand','head','cluster.agg')})"
done
