run() { SAD_AMD_LIB=$LIB timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$LIB] $*', d['value'], d['ms_per_step'], d['step_ms']['p50'])"; }
for i in 1 2 3; do
LIB=build/libsad_bqprev.so run
LIB= run
done
LIB=build/libsad_bqprev.so run --dtype bf16
LIB= run --dtype bf16
LIB=build/libsad_bqprev.so run --dtype bf16
LIB= run --dtype bf16
SAD_AMD_LIB=build/libsad_bqprev.so python tools/bq_bench.py 2>/dev/null | sed "s/^/prev: /"; python tools/bq_bench.py 2>/dev/null | sed "s/^/new:  /"
