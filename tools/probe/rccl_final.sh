#!/bin/bash
# the one-rank RCCL rehearsal with the distributed stream set (six sampling streams, no ingest stream) against the plain run, and
# k further idle queues on top (how much room the set leaves the backend)
set -eo pipefail
out=gpurun_out/${1:-rcclf}
mkdir -p $out
C="--no-legs --no-cpu --no-dense-leg --no-launch-timing"
python bench.py --save-geometry $out/g32.json $C --steps 20 --warmup 5 > $out/t32.json 2> $out/t32.err
for r in 1 2; do
  python bench.py --geometry-file $out/g32.json $C --steps 200 --warmup 16 > $out/plain_$r.json 2> $out/plain_$r.err
  python bench.py --geometry-file $out/g32.json $C --steps 20 --warmup 5 > $out/plain20_$r.json 2> $out/plain20_$r.err
  for k in 0 4 6 8; do
    SAD_BENCH_FORCE_DIST=1 SAD_EXTRA_QUEUES=$k python tools/probe/bench_extra_queues.py --geometry-file $out/g32.json $C --steps 200 --warmup 16 > $out/rccl_k${k}_$r.json 2> $out/rccl_k${k}_$r.err
  done
  SAD_BENCH_FORCE_DIST=1 python bench.py --geometry-file $out/g32.json $C --steps 20 --warmup 5 > $out/rccl20_$r.json 2> $out/rccl20_$r.err
done
python - $out <<'P'
import json, glob, sys, re
rows = {}
for f in sorted(glob.glob(f"{sys.argv[1]}/*_[12].json")):
    t = open(f).read().strip()
    k = re.sub(r"_\d+\.json$", "", f.split("/")[-1])
    d = json.loads(t.splitlines()[-1]) if t else None
    rows.setdefault(k, []).append((d["value"], d["config"]["fps_streams"]) if d else None)
for k, v in rows.items():
    print(k, v)
P
