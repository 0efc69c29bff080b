#!/bin/bash
# NOTE: under rocprofv3 the host needs ~1.2 ms per submit, so the 0.66 ms bf16 step becomes host-bound (2.0 ms/step in the trace): the per-queue
# view is only meaningful for the f32 step (2.1 ms) or with --steps large and the trace read for kernel durations alone.
set -eo pipefail
out=gpurun_out/${1:-trbf}
mkdir -p $out
C="--no-legs --no-cpu --no-dense-leg --no-launch-timing --no-bf16-quality"
python bench.py --dtype bf16 --save-geometry $out/g16.json $C --steps 20 --warmup 16 > $out/t16.json 2> $out/t16.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$out/tr -- python3 $R/bench.py --dtype bf16 --geometry-file $R/$out/g16.json $C --steps 300 --warmup 16 > $R/$out/b.json 2> $R/$out/b.err
cd $R
python tools/probe/trace_queues.py $out/tr
