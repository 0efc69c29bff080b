#!/bin/bash
# stream placement (sad_amd._runtime.placed_streams) on / off, alternating default runs with all legs; then the one-rank RCCL
# rehearsal on / off with placement
set -eo pipefail
out=gpurun_out/${1:-place1}
mkdir -p $out
python bench.py --save-geometry $out/geom.json --no-legs --no-cpu --steps 20 --warmup 5 > $out/tune.json 2> $out/tune.err
A="--geometry-file $out/geom.json --no-cpu --no-dense-leg"
for r in 1 2 3; do
  SAD_NO_STREAM_PLACEMENT=1 python bench.py $A > $out/off_$r.json 2> $out/off_$r.err
  python bench.py $A > $out/on_$r.json 2> $out/on_$r.err
  SAD_BENCH_FORCE_DIST=1 python bench.py $A --no-legs > $out/onrccl_$r.json 2> $out/onrccl_$r.err
  python bench.py $A --no-legs --steps 20 --warmup 5 > $out/on20_$r.json 2> $out/on20_$r.err
  SAD_NO_STREAM_PLACEMENT=1 python bench.py $A --no-legs --steps 20 --warmup 5 > $out/off20_$r.json 2> $out/off20_$r.err
done
python - $out <<'P'
import json, glob, sys
for mode in ("off", "on", "onrccl", "off20", "on20"):
    for f in sorted(glob.glob(f"{sys.argv[1]}/{mode}_*.json")):
        t = open(f).read().strip()
        if not t:
            print(mode, "FAILED", open(f.replace(".json", ".err")).read()[-300:]); continue
        d = json.loads(t.splitlines()[-1])
        print(mode, d["value"], *[(k, d[k].get("value")) for k in ("bf16_leg", "configs4_leg", "pipeline_leg") if k in d], d["parity_check"]["ok"])
P
