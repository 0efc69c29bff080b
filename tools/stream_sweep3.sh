for ms in 2 3 4; do for fs in 2 3; do
  python3 bench.py --no-cpu --no-dense-leg --no-launch-timing --steps 300 --main-streams $ms --fps-streams $fs --geometry-file profiles/r02_geometry.json 2>/dev/null | grep -E '^\{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('main $ms fps $fs:', j['value'], j['ms_per_step'])"
done; done
