#!/bin/bash
# kernel traces of the pipelined step without and with the gather's stream pattern (no collective): where do the 5 % go?
set -eo pipefail
out=gpurun_out/${1:-trg}
mkdir -p $out
python bench.py --save-geometry $out/geom.json --no-legs --no-cpu --steps 20 --warmup 5 > $out/tune.json 2> $out/tune.err
R=$PWD
A="--geometry-file $R/$out/geom.json --no-legs --no-cpu --no-dense-leg --no-launch-timing --steps 100 --warmup 14"
cd /tmp && export TMPDIR=/tmp
for m in none full; do
  SAD_FAKE3=$m rocprofv3 --kernel-trace --output-format csv -d $R/$out/$m -- python3 $R/tools/probe/bench_fake_gather3.py $A > $R/$out/$m.json 2> $R/$out/$m.err
done
cd $R
python tools/probe/trace_gather_modes.py $out
