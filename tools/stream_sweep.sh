# throughput vs stream configuration (geometry from a file: identical kernels in every run)
python3 bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 100 --save-geometry gpurun_out/geom_sweep.json > gpurun_out/ss_base.log 2>&1
grep -E '^\{' gpurun_out/ss_base.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('autotuned base', j['value'])"
for cfg in "1 4" "2 4" "3 4" "2 2" "2 6" "4 4"; do set -- $cfg
  python3 bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 100 --main-streams $1 --fps-streams $2 --geometry-file gpurun_out/geom_sweep.json 2>/dev/null | grep -E '^\{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('main $1 fps $2:', j['value'], j['ms_per_step'])"
done
