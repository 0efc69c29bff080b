#!/bin/bash
set -eo pipefail
out=gpurun_out/${1:-place4}
mkdir -p $out
C="--no-legs --no-cpu --no-dense-leg --no-launch-timing --no-bf16-quality"
python bench.py --save-geometry $out/g32.json $C --steps 20 --warmup 5 > $out/t32.json 2> $out/t32.err
python bench.py --dtype bf16 --save-geometry $out/g16.json $C --steps 20 --warmup 14 > $out/t16.json 2> $out/t16.err
export GPU_MAX_HW_QUEUES=32
for r in 1 2 3; do
  for f in 8 10 12; do
    python bench.py --geometry-file $out/g32.json $C --fps-streams $f --steps 20 --warmup 5 > $out/f32_f${f}_$r.json 2> $out/f32_f${f}_$r.err
    python bench.py --geometry-file $out/g32.json $C --fps-streams $f --steps 200 --warmup 14 > $out/l32_f${f}_$r.json 2> $out/l32_f${f}_$r.err
    python bench.py --dtype bf16 --geometry-file $out/g16.json $C --fps-streams $f --steps 300 --warmup 14 > $out/bf16_f${f}_$r.json 2> $out/bf16_f${f}_$r.err
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
