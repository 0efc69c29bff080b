# sampling-stream sweep for the three workloads (after the round-4 FPS kernels): bash tools/probe/stream_sweep.sh
run() { timeout -k 10 200 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['step_ms']['p50'])"; }
for n in 2 3 4; do run --fps-streams $n --steps 200; done
for n in 4 6 8; do run --dtype bf16 --fps-streams $n --steps 200; done
for n in 4 5 6 8; do run --config nuscenes --dtype bf16 --fps-streams $n --steps 80 --batches 2; done
