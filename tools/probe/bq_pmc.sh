# instruction mix of the grid ball query (SA1/SA2 shapes): two counter passes over tools/bq_bench.py
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/bq_pmc
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/a -- python3 tools/bq_bench.py > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $out/b -- python3 tools/bq_bench.py > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/bq_pmc")
for tag in "ab":
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "grid_query" not in k: continue
            k += " grid=" + r.get("Grid_Size", "?")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()): print(f"   {c:24s} {v / n[(k, c)]:16.0f} per launch")
PY
