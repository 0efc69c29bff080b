#!/bin/bash
# One-rank RCCL rehearsal of the N > 1 path on a one-GPU box: the GPU dist tests, then the bench at the driver's setting and at
# 200 steps with and without the forced collective (alternating), into gpurun_out/$1.
set -eo pipefail
out=gpurun_out/${1:-rccl1}
mkdir -p $out
python -m pytest tests/test_gpu_dist.py -x -q -m gpu > $out/tests.log 2>&1
python bench.py --save-geometry $out/geom.json --no-legs --no-cpu --steps 20 --warmup 5 > $out/tune.json 2> $out/tune.err
for r in 1 2 3; do
  for mode in plain rccl; do
    for st in "20 5" "200 14"; do
      set -- $st
      if [ $mode = rccl ]; then export SAD_BENCH_FORCE_DIST=1; else unset SAD_BENCH_FORCE_DIST; fi
      python bench.py --geometry-file $out/geom.json --no-legs --no-cpu --steps $1 --warmup $2 > $out/${mode}_$1_$r.json 2> $out/${mode}_$1_$r.err
    done
  done
done
unset SAD_BENCH_FORCE_DIST
python - $out <<'P'
import json, glob, sys, os
for mode in ("plain", "rccl"):
    for st in (20, 200):
        v = []
        for f in sorted(glob.glob(f"{sys.argv[1]}/{mode}_{st}_*.json")):
            d = json.loads(open(f).read().strip().splitlines()[-1])
            v.append(d["value"])
            rk = d.get("ranks")
        print(mode, st, v, "ranks:", {k: rk[k] for k in ("backend", "ranks_seen", "one_rank_rehearsal")} if rk else None)
P
