# Round-end evidence run: full GPU test suite, the bench line, rocprofv3 kernel statistics (three-stream and single-stream backward)
# and the PMC passes.  Everything lands under gpurun_out/r2_final/ ; tools/collect_profiles.py condenses it into profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_final
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -12 $O/gpu_tests.log
python bench.py > $O/bench_line.json 2> $O/bench_line.err
cat $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_ss -o ss --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof_ss.err
MMNN_SIDE_STREAMS=2 rocprofv3 --kernel-trace --stats -d $O/prof_ov -o ov --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_two_side_streams.json 2> $O/prof_ov.err
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$name -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_$name.log 2>&1 || echo "pass $name failed"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM
find $O -name "*kernel_trace.csv" -delete
du -sh $O
