# Pipelined-step A/B of several libraries inside ONE gpurun call (same box, alternating): usage
#   bash tools/probe/ab_r05.sh OUTDIR ROUNDS "bench args" lib1 lib2 ...     (lib = name under build/libsad_<name>.so, "tree" = the in-tree build)
# The first run autotunes and saves the geometry; every later run loads it (no autotune launches, identical dispatch kernels).
out=$1; rounds=$2; bargs=$3; shift 3
mkdir -p $out
geom=$out/geometry.json
if [ ! -f $geom ]; then
  timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs --steps 50 --warmup 5 $bargs --save-geometry $geom > $out/tune.json 2> $out/tune.err || { echo "tune run failed"; tail -5 $out/tune.err; exit 1; }
fi
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    L=""; [ $lib != tree ] && L=build/libsad_$lib.so
    SAD_BENCH_WHATIF=1 SAD_AMD_LIB=$L timeout -k 10 300 python bench.py --no-cpu --no-dense-leg --no-launch-timing --no-legs --steps 200 --warmup 10 $bargs --geometry-file $geom > $out/${lib}_$r.json 2> $out/${lib}_$r.err
    python - <<PY
import json
try:
    d = json.loads(open("$out/${lib}_$r.json").read().strip().splitlines()[-1])
    print(f"$lib round $r: {d['value']:9.1f} scenes/s  {d['ms_per_step']:.4f} ms/step  p50 {d['step_ms']['p50']:.4f}  over2x {d['step_ms']['over_2x_p50']}", flush=True)
except Exception as e:
    print("$lib round $r: FAILED", e, flush=True)
PY
  done
done
python - "$out" "$@" <<'PY'
import json, sys, glob, statistics
out, libs = sys.argv[1], sys.argv[2:]
print("summary (mean ms/step over rounds; delta vs the first library):")
base = None
for lib in libs:
    v = []
    for f in sorted(glob.glob(f"{out}/{lib}_*.json")):
        try:
            v.append(json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"])
        except Exception:
            pass
    if not v:
        continue
    m = statistics.mean(v)
    base = m if base is None else base
    print(f"  {lib:12s} {m:.4f} ms  (min {min(v):.4f} max {max(v):.4f}, n={len(v)})  {100 * (m - base) / base:+.2f} %")
PY
