#!/bin/bash
# what the one-rank RCCL collective costs and why: plain / RCCL / same stream pattern with a device copy, alternating; then a
# kernel trace of the RCCL mode
set -eo pipefail
out=gpurun_out/${1:-rccl2}
mkdir -p $out
python bench.py --save-geometry $out/geom.json --no-legs --no-cpu --steps 20 --warmup 5 > $out/tune.json 2> $out/tune.err
for r in 1 2 3; do
  unset SAD_BENCH_FORCE_DIST
  python bench.py --geometry-file $out/geom.json --no-legs --no-cpu --no-dense-leg --steps 200 --warmup 14 > $out/plain_$r.json 2> $out/plain_$r.err
  export SAD_BENCH_FORCE_DIST=1
  python bench.py --geometry-file $out/geom.json --no-legs --no-cpu --no-dense-leg --steps 200 --warmup 14 > $out/rccl_$r.json 2> $out/rccl_$r.err
  python tools/probe/bench_fake_gather.py --geometry-file $out/geom.json --no-legs --no-cpu --no-dense-leg --steps 200 --warmup 14 > $out/copy_$r.json 2> $out/copy_$r.err
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$out/prof -o rccl -- python3 $R/bench.py --geometry-file $R/$out/geom.json --no-legs --no-cpu --no-dense-leg --no-launch-timing --steps 60 --warmup 14 > $R/$out/prof.json 2> $R/$out/prof.err
cd $R
python - $out <<'P'
import json, glob, sys
for mode in ("plain", "rccl", "copy"):
    v = []
    for f in sorted(glob.glob(f"{sys.argv[1]}/{mode}_*.json")):
        t = open(f).read().strip()
        if t: v.append(json.loads(t.splitlines()[-1])["value"])
    print(mode, v)
P
find $out/prof -name "*kernel_stats.csv" | while read f; do cut -c1-150 $f | sed -n 1,14p; done
