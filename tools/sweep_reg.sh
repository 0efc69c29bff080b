set -e
for spec in "sa1.b0 24802 2" "sa1.b1 25802 2" "sa1.b2 26801 2" "sa2.b0 25811 2" "sa2.b1 25811 2" "sa2.b2 126821 2" "sa3.b0 25832 2" "sa3.b1 25831 2" "sa3.b2 24831 2"; do
  python tools/mlp_real_sweep.py $spec 2>&1 | grep -v amdgpu.ids
done
