# rocprofv3 kernel-trace summary + the bench line of the same command (usage: bash tools/profile.sh v9 [bench flags])
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu "$@" > gpurun_out/prof_${tag}_bench.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_bench.log; exit 1; }
grep -E '^\{' gpurun_out/prof_${tag}_bench.log | cut -c1-160
find gpurun_out/prof_$tag -name "*kernel_stats.csv"
