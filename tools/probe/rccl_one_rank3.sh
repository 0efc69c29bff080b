#!/bin/bash
set -eo pipefail
out=gpurun_out/${1:-rccl3}
mkdir -p $out
python bench.py --save-geometry $out/geom.json --no-legs --no-cpu --steps 20 --warmup 5 > $out/tune.json 2> $out/tune.err
A="--geometry-file $out/geom.json --no-legs --no-cpu --no-dense-leg --no-launch-timing --steps 200 --warmup 14"
for r in 1 2; do
  unset SAD_BENCH_FORCE_DIST
  python bench.py $A > $out/plain_$r.json 2> $out/plain_$r.err
  export SAD_BENCH_FORCE_DIST=1
  python bench.py $A > $out/rccl_$r.json 2> $out/rccl_$r.err
  for m in memcpy kernel events main; do
    SAD_FAKE=$m python tools/probe/bench_fake_gather.py $A > $out/${m}_$r.json 2> $out/${m}_$r.err
  done
done
python - $out <<'P'
import json, glob, sys
for mode in ("plain", "rccl", "memcpy", "kernel", "events", "main"):
    v = []
    for f in sorted(glob.glob(f"{sys.argv[1]}/{mode}_*.json")):
        t = open(f).read().strip()
        if t: v.append(json.loads(t.splitlines()[-1])["value"])
    print(mode, v)
P
