# PMC passes for the bench (separate passes per counter group; kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
run() { # name counters...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$n -- python3 bench.py --no-cpu --no-dense-leg --geometry-file profiles/r02_geometry.json --no-launch-timing --main-streams 1 --steps 3 --warmup 1 > gpurun_out/pmc/$n.log 2>&1 || { echo "pass $n failed"; tail -5 gpurun_out/pmc/$n.log; return 1; }
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
find gpurun_out/pmc -name "*counter_collection.csv" | head
