set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "backbone_forward_backward or tile_matrix or repeated" > gpurun_out/r2_t7.log 2>&1 || { tail -40 gpurun_out/r2_t7.log; exit 1; }
tail -3 gpurun_out/r2_t7.log
python tools/exp_classes.py - -
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MMNN_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r2_pmc_fetch2 -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/r2_pmc_fetch2.log 2>&1
find $R/gpurun_out/r2_pmc_fetch2 -name "*kernel_trace.csv" -delete
cd $R
TOP=14 python tools/pmc_summary.py gpurun_out/r2_pmc_fetch2/*counter_collection.csv | cut -c1-120
