#!/bin/bash
set -eo pipefail
out=gpurun_out/${1:-place3}
mkdir -p $out
C="--no-legs --no-cpu --no-dense-leg --no-launch-timing"
python bench.py --save-geometry $out/g32.json $C --steps 20 --warmup 5 > $out/t32.json 2> $out/t32.err
for r in 1 2 3 4; do
  for f in 2 3 4 6 8; do
    python bench.py --geometry-file $out/g32.json $C --fps-streams $f --steps 20 --warmup 5 > $out/f32_f${f}_$r.json 2> $out/f32_f${f}_$r.err
  done
done
for r in 1 2; do
  for f in 4 6 8; do
    python bench.py --geometry-file $out/g32.json $C --fps-streams $f --steps 200 --warmup 14 > $out/l32_f${f}_$r.json 2> $out/l32_f${f}_$r.err
  done
done
python - $out <<'P'
import json, glob, sys, re
rows = {}
for f in sorted(glob.glob(f"{sys.argv[1]}/*_f*_*.json")):
    t = open(f).read().strip()
    k = re.sub(r"_\d+\.json$", "", f.split("/")[-1])
    rows.setdefault(k, []).append(json.loads(t.splitlines()[-1])["value"] if t else None)
for k, v in rows.items():
    print(k, v)
P
