#!/bin/bash
set -eo pipefail
out=gpurun_out/${1:-minst}
mkdir -p $out
python bench.py --save-geometry $out/g32.json --no-legs --no-cpu --steps 20 --warmup 5 > $out/t32.json 2> $out/t32.err
C="--geometry-file $out/g32.json --no-legs --no-cpu --no-dense-leg"
for r in 1 2 3; do
  python bench.py $C > $out/all8_$r.json 2> $out/all8_$r.err
  python tools/probe/bench_min_streams.py $C --fps-streams 3 > $out/min3_$r.json 2> $out/min3_$r.err
  python tools/probe/bench_min_streams.py $C --fps-streams 8 > $out/min8_$r.json 2> $out/min8_$r.err
  python bench.py $C --fps-streams 3 > $out/all3_$r.json 2> $out/all3_$r.err
done
python - $out <<'P'
import json, glob, sys
for mode in ("all8", "min8", "all3", "min3"):
    for f in sorted(glob.glob(f"{sys.argv[1]}/{mode}_*.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(mode, d["value"], "frac", r["frac"], "in-step ms", r["ms_per_step"], r["ms_per_step_uncorrected"], "pair", r["event_pair_ms"], "steady", r["steady_clock"]["frac"])
P
