#!/bin/bash
set -eo pipefail
out=gpurun_out/${1:-m3}
mkdir -p $out
C="--no-legs --no-cpu --no-dense-leg --no-launch-timing --no-bf16-quality"
python bench.py --save-geometry $out/g32.json $C --steps 20 --warmup 5 > $out/t32.json 2> $out/t32.err
python bench.py --dtype bf16 --save-geometry $out/g16.json $C --steps 20 --warmup 16 > $out/t16.json 2> $out/t16.err
for r in 1 2; do
  python bench.py --geometry-file $out/g32.json $C --steps 200 --warmup 16 > $out/f32_m2f8_$r.json 2> $out/f32_m2f8_$r.err
  python bench.py --dtype bf16 --geometry-file $out/g16.json $C --steps 300 --warmup 16 > $out/bf16_m2f8_$r.json 2> $out/bf16_m2f8_$r.err
  for f in 3 4 5; do
    python bench.py --geometry-file $out/g32.json $C --main-streams 3 --fps-streams $f --steps 200 --warmup 16 > $out/f32_m3f${f}_$r.json 2> $out/f32_m3f${f}_$r.err
  done
  for f in 4 5; do
    python bench.py --dtype bf16 --geometry-file $out/g16.json $C --main-streams 3 --fps-streams $f --steps 300 --warmup 16 > $out/bf16_m3f${f}_$r.json 2> $out/bf16_m3f${f}_$r.err
  done
done
python - $out <<'P'
import json, glob, sys, re
rows = {}
for f in sorted(glob.glob(f"{sys.argv[1]}/*_m*_[12].json")):
    t = open(f).read().strip()
    k = re.sub(r"_\d+\.json$", "", f.split("/")[-1])
    rows.setdefault(k, []).append(json.loads(t.splitlines()[-1])["value"] if t else open(f.replace(".json",".err")).read()[-200:])
for k, v in rows.items():
    print(k, v)
P
