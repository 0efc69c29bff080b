# A/B of sad_set_option knobs on the whole pipeline: bash tools/ab_bench.sh "mlp_static=0" "mlp_static=1" ...
for o in "$@"; do
  for rep in 1 2; do
  python3 bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 300 --opt $o --geometry-file profiles/r02_geometry.json 2>/dev/null | grep -E '^\{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$o:', j['value'], j['ms_per_step'])"
  done
done
