# pipelined step, f32 and bf16: the library named by $1 against the build in the tree, three rounds
run() { SAD_AMD_LIB=$LIB timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$LIB] $*', d['value'], d['ms_per_step'], d['step_ms']['p50'])"; }
for i in 1 2 3; do LIB=$1 run; LIB= run; done
for i in 1 2; do LIB=$1 run --dtype bf16; LIB= run --dtype bf16; done
