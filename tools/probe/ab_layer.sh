# pipelined step: committed layer kernel vs this build (static deal) vs this build with mlp_layer_queue=1
for i in 1 2; do
for v in "build/libsad_layerold.so|" "|" "|--opt mlp_layer_queue=1"; do
  lib=${v%%|*}; opt=${v##*|}
  SAD_AMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 200 --warmup 10 $opt 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=[$lib] opt=[$opt]', d['value'], d['ms_per_step'], d['step_ms']['p50'])"
done; done
