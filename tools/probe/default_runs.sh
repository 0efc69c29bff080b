#!/bin/bash
# the default run and the driver's command, alternating, n times: gpurun_out/$1
set -eo pipefail
out=gpurun_out/${1:-runs}
n=${2:-3}
mkdir -p $out
for r in $(seq 1 $n); do
  python bench.py > $out/default_$r.json 2> $out/default_$r.err
  python bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver_$r.json 2> $out/driver_$r.err
done
python - $out <<'P'
import json, glob, sys
for mode in ("default", "driver"):
    for f in sorted(glob.glob(f"{sys.argv[1]}/{mode}_*.json")):
        t = open(f).read().strip()
        if not t:
            print(mode, "FAILED", open(f.replace(".json", ".err")).read()[-300:]); continue
        d = json.loads(t.splitlines()[-1])
        print(mode, d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], *[(k, d[k].get("value")) for k in ("bf16_leg", "configs4_leg", "pipeline_leg") if k in d],
              "parity", d["parity_check"]["ok"], "cpu", d.get("cpu_baseline", {}).get("value"))
P
