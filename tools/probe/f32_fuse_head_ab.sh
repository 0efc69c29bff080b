# f32: cluster.agg + head as ONE fused chain (default) against cluster.agg on the row-streaming layer + head as its own chain (SAD_F32_NO_FUSE_HEAD=1),
# alternating pipelined runs, each with its own autotune pass:  bash tools/probe/f32_fuse_head_ab.sh ROUNDS
for r in $(seq 1 ${1:-3}); do
  for m in fused split; do
    E=""; [ $m = split ] && E=1
    SAD_F32_NO_FUSE_HEAD=$E timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs --steps 200 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m round $r:', d['value'], d['ms_per_step'], d.get('parity_check',{}).get('ok'), {k:v for k,v in d['config']['mlp_geometry'].items() if 'agg' in k or 'head' in k})"
  done
done
