for cfg in "2 1" "2 2" "2 3" "1 2" "3 2"; do set -- $cfg
  python3 bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 100 --main-streams $1 --fps-streams $2 --geometry-file profiles/r02_geometry.json 2>/dev/null | grep -E '^\{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('main $1 fps $2:', j['value'], j['ms_per_step'])"
done
