# per-kernel durations of a short serial run (one stream, nothing overlapped): bash tools/probe/serial_stats.sh <pattern> [bench args]
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/serial_stats; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu --no-dense-leg --no-launch-timing --main-streams 1 --no-overlap --steps 10 --warmup 2 "$@" > $out/log.txt 2>&1 || { tail -3 $out/log.txt; exit 1; }
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(p in r["Name"] for p in sys.argv[2].split(",")):
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us min {float(r["MinNs"])/1e3:8.1f}')
PY
