# PMC passes around one python tool run (kernel-trace only, as gpurun requires).
# usage: bash tools/pmc_one.sh <outdir-name> <python script and args...>
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$name
run() { # pass counters...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/$name/$n -- python3 "${CMD[@]}" > gpurun_out/$name/$n.log 2>&1 || { echo "pass $n failed"; tail -5 gpurun_out/$name/$n.log; return 1; }
}
CMD=("$@")
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 &&
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
python3 tools/pmc_summary.py gpurun_out/$name > gpurun_out/$name/summary.txt && grep -A30 "mlp_reg\|mlp_chain_kernel\|mlp_multi" gpurun_out/$name/summary.txt | head -80
