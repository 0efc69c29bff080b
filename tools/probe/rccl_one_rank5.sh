#!/bin/bash
set -eo pipefail
out=gpurun_out/${1:-rccl5}
mkdir -p $out
python bench.py --save-geometry $out/geom.json --no-legs --no-cpu --steps 20 --warmup 5 > $out/tune.json 2> $out/tune.err
A="--geometry-file $out/geom.json --no-legs --no-cpu --no-dense-leg --no-launch-timing --steps 200 --warmup 14"
for r in 1 2; do
  for m in none full sidewait onside late; do
    SAD_FAKE3=$m python tools/probe/bench_fake_gather3.py $A > $out/${m}_$r.json 2> $out/${m}_$r.err
  done
done
python - $out <<'P'
import json, glob, sys
for mode in ("none", "full", "sidewait", "onside", "late"):
    v = []
    for f in sorted(glob.glob(f"{sys.argv[1]}/{mode}_*.json")):
        t = open(f).read().strip()
        h = [l for l in open(f.replace(".json", ".err")).read().splitlines() if l.startswith("HOST")]
        v.append((json.loads(t.splitlines()[-1])["value"] if t else None, h[-1] if h else None))
    print(mode, v)
P
