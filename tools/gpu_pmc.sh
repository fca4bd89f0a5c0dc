# PMC passes of the bench command (separate passes per counter group, kernel trace only).  Output: gpurun_out/r2_pmc_*/...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/r2_counters_list.txt 2>&1 || true
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/r2_pmc_$name -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/r2_pmc_$name.log 2>&1 || echo "pass $name failed"
  find $R/gpurun_out/r2_pmc_$name -name "*kernel_trace.csv" -delete
}
export MMNN_SINGLE_STREAM=1
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM
cd $R
ls gpurun_out/r2_pmc_* | head -30
for p in fetch write sq1 sq2; do echo "== $p"; TOP=12 python tools/pmc_summary.py gpurun_out/r2_pmc_$p/*counter_collection.csv 2>&1 | cut -c1-330 | head -16; done
