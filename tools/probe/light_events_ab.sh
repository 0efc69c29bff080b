#!/bin/bash
set -eo pipefail
out=gpurun_out/${1:-light1}
mkdir -p $out
python bench.py --save-geometry $out/geom.json --no-legs --no-cpu --steps 20 --warmup 5 > $out/tune.json 2> $out/tune.err
A="--geometry-file $out/geom.json --no-legs --no-cpu --no-dense-leg --no-launch-timing --steps 200 --warmup 14"
for r in 1 2 3; do
  python bench.py $A > $out/plain_$r.json 2> $out/plain_$r.err
  SAD_LIGHT=0 python tools/probe/bench_light_events.py $A > $out/ctl_$r.json 2> $out/ctl_$r.err
  SAD_LIGHT=1 python tools/probe/bench_light_events.py $A > $out/light_$r.json 2> $out/light_$r.err
  SAD_LIGHT=0 SAD_FAKE2=full python tools/probe/bench_light_events.py $A > $out/ctlg_$r.json 2> $out/ctlg_$r.err
  SAD_LIGHT=1 SAD_FAKE2=full python tools/probe/bench_light_events.py $A > $out/lightg_$r.json 2> $out/lightg_$r.err
done
python - $out <<'P'
import json, glob, sys
for mode in ("plain", "ctl", "light", "ctlg", "lightg"):
    v = []
    for f in sorted(glob.glob(f"{sys.argv[1]}/{mode}_*.json")):
        t = open(f).read().strip()
        if t:
            d = json.loads(t.splitlines()[-1]); v.append((d["value"], d["parity_check"]["ok"]))
        else:
            v.append(open(f.replace(".json", ".err")).read()[-300:])
    print(mode, v)
P
