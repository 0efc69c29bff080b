# per-branch timings of the SA stages with the given geometry codes: bash tools/sa_sweep.sh 2 4
for n in sa1.b0 sa1.b2 sa2.b0 sa2.b2 sa3.b0 sa3.b1 sa3.b2 cluster.b0; do python3 tools/mlp_real_sweep.py $n "$@" | tr '\n' ' '; echo; done
