#!/bin/bash
# FPS from sorted records in L2 (fps_variant = 5: fps_cellg2_kernel<16>, half the register footprint of the register-resident kernel,
# slower per step) against the default, pipelined runs: does a CU that runs an FPS workgroup become usable for the MLP kernels?
set -eo pipefail
out=gpurun_out/${1:-fpsrec}
mkdir -p $out
C="--no-legs --no-cpu --no-dense-leg --no-launch-timing --no-bf16-quality"
python bench.py --save-geometry $out/g32.json $C --steps 20 --warmup 5 > $out/t32.json 2> $out/t32.err
python bench.py --dtype bf16 --save-geometry $out/g16.json $C --steps 20 --warmup 16 > $out/t16.json 2> $out/t16.err
for r in 1 2 3; do
  for v in 0 5; do
    python bench.py --geometry-file $out/g32.json $C --opt fps_variant=$v --steps 200 --warmup 16 > $out/f32_v${v}_$r.json 2> $out/f32_v${v}_$r.err
    python bench.py --dtype bf16 --geometry-file $out/g16.json $C --opt fps_variant=$v --steps 300 --warmup 16 > $out/bf16_v${v}_$r.json 2> $out/bf16_v${v}_$r.err
  done
done
python - $out <<'P'
import json, glob, sys, re
rows = {}
for f in sorted(glob.glob(f"{sys.argv[1]}/*_v*_*.json")):
    t = open(f).read().strip()
    k = re.sub(r"_\d+\.json$", "", f.split("/")[-1])
    rows.setdefault(k, []).append(json.loads(t.splitlines()[-1])["value"] if t else open(f.replace(".json",".err")).read()[-200:])
for k, v in rows.items():
    print(k, v)
P
