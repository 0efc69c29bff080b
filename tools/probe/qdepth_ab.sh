#!/bin/bash
set -eo pipefail
out=gpurun_out/${1:-qd}
mkdir -p $out
C="--no-legs --no-cpu --no-dense-leg --no-launch-timing --no-bf16-quality"
python bench.py --save-geometry $out/g32.json $C --steps 20 --warmup 5 > $out/t32.json 2> $out/t32.err
python bench.py --dtype bf16 --save-geometry $out/g16.json $C --steps 20 --warmup 16 > $out/t16.json 2> $out/t16.err
for r in 1 2; do
  for q in 4 6 10 14; do
    python bench.py --geometry-file $out/g32.json $C --queue-depth $q --steps 200 --warmup 16 > $out/f32_q${q}_$r.json 2> $out/f32_q${q}_$r.err
    python bench.py --geometry-file $out/g32.json $C --queue-depth $q --steps 20 --warmup 5 > $out/d32_q${q}_$r.json 2> $out/d32_q${q}_$r.err
    python bench.py --dtype bf16 --geometry-file $out/g16.json $C --queue-depth $q --steps 300 --warmup 16 > $out/bf16_q${q}_$r.json 2> $out/bf16_q${q}_$r.err
  done
done
python - $out <<'P'
import json, glob, sys, re
rows = {}
for f in sorted(glob.glob(f"{sys.argv[1]}/*_q*_*.json")):
    t = open(f).read().strip()
    k = re.sub(r"_\d+\.json$", "", f.split("/")[-1])
    rows.setdefault(k, []).append(json.loads(t.splitlines()[-1])["value"] if t else None)
for k, v in rows.items():
    print(k, v)
P
