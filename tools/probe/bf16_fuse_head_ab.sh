# bf16 KITTI-shaped leg with cluster.agg and head as two launches (default) against ONE fused four-layer chain on the tiled bf16 kernel
for r in 1 2 3; do for v in "" 1; do
SAD_BF16_FUSE_HEAD=$v timeout -k 10 300 python bench.py --dtype bf16 --no-cpu --no-legs --no-dense-leg --no-launch-timing --no-bf16-quality --steps 400 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused' if '$v' else 'two launches', d['value'], d['ms_per_step'], d['step_ms']['p50'])"
done; done
