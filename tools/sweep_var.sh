for c in sa1.b0 sa1.b2 sa2.b0 sa2.b2 sa3.b0 sa3.b1 sa3.b2; do python tools/mlp_real_sweep.py $c 2 2>&1 | grep -E "^sa| 2:" | tr '\n' ' '; echo; done
