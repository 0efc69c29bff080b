mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/t3.log 2>&1; echo exit=$? >> gpurun_out/t3.log; tail -3 gpurun_out/t3.log
for o in "--main-streams 1" "--no-launch-timing" "--no-launch-timing"; do
  timeout -k 10 200 python bench.py --no-cpu $o > gpurun_out/b2.tmp 2>&1; grep -E '^\{' gpurun_out/b2.tmp >> gpurun_out/b2.log || tail -5 gpurun_out/b2.tmp
done
python - <<'PY'
import json
for l in open('gpurun_out/b2.log'):
    d=json.loads(l)
    if 'roofline' not in d: print('no-timing', d['value'], d['ms_per_step']); continue
    print(d['config']['opts'], d['value'], d['ms_per_step'], 'mlp', d['roofline']['ms_per_step'], d['roofline']['achieved'], 'fps', d['kernels'][0]['ms_per_step'], 'bq', d['kernels'][1]['ms_per_step']); print('   ', {k:(v['ms'],v['executed_gflop'],v['tflops']) for k,v in d['mlp_launches'].items()})
PY
