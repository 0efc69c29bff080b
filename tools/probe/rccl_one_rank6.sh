#!/bin/bash
set -eo pipefail
out=gpurun_out/${1:-rccl6}
mkdir -p $out
python bench.py --save-geometry $out/geom.json --no-legs --no-cpu --steps 20 --warmup 5 > $out/tune.json 2> $out/tune.err
A="--geometry-file $out/geom.json --no-legs --no-cpu --no-dense-leg --no-launch-timing --steps 200 --warmup 14"
for r in 1 2; do
  SAD_FAKE3=none python tools/probe/bench_fake_gather3.py $A > $out/none_$r.json 2> $out/none_$r.err
  for k in 0 1 2 3 4 5; do
    SAD_DUMMY=$k SAD_FAKE3=full python tools/probe/bench_fake_gather3.py $A > $out/d${k}_$r.json 2> $out/d${k}_$r.err
  done
done
python - $out <<'P'
import json, glob, sys
for mode in ("none", "d0", "d1", "d2", "d3", "d4", "d5"):
    v = []
    for f in sorted(glob.glob(f"{sys.argv[1]}/{mode}_*.json")):
        t = open(f).read().strip()
        v.append(json.loads(t.splitlines()[-1])["value"] if t else open(f.replace(".json", ".err")).read()[-200:])
    print(mode, v)
P
