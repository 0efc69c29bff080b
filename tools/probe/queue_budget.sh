#!/bin/bash
set -eo pipefail
out=gpurun_out/${1:-qbud}
mkdir -p $out
C="--no-legs --no-cpu --no-dense-leg --no-launch-timing --no-bf16-quality"
python bench.py --save-geometry $out/g32.json $C --steps 20 --warmup 5 > $out/t32.json 2> $out/t32.err
export GPU_MAX_HW_QUEUES=32
for r in 1 2; do
  for k in 0 1 2 3 4 5 6 8; do
    SAD_EXTRA_QUEUES=$k python tools/probe/bench_extra_queues.py --geometry-file $out/g32.json $C --steps 200 --warmup 16 > $out/f32_k${k}_$r.json 2> $out/f32_k${k}_$r.err
  done
done
python - $out <<'P'
import json, glob, sys, re
rows = {}
for f in sorted(glob.glob(f"{sys.argv[1]}/*_k*_*.json")):
    t = open(f).read().strip()
    k = re.sub(r"_\d+\.json$", "", f.split("/")[-1])
    rows.setdefault(k, []).append(json.loads(t.splitlines()[-1])["value"] if t else None)
for k, v in rows.items():
    print(k, v)
P
