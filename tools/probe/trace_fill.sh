#!/bin/bash
# kernel trace of the driver's setting: what happens between the start of the timed region and the first MLP kernels
set -eo pipefail
out=gpurun_out/${1:-fill}
mkdir -p $out
python bench.py --save-geometry $out/geom.json --no-legs --no-cpu --steps 20 --warmup 5 > $out/tune.json 2> $out/tune.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$out/tr -- python3 $R/bench.py --geometry-file $R/$out/geom.json --no-legs --no-cpu --no-dense-leg --no-launch-timing --steps 20 --warmup 5 > $R/$out/b.json 2> $R/$out/b.err
cd $R
python tools/probe/trace_fill.py $out
