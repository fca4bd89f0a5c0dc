set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/r2_trace -o tr --output-format csv -- python3 $R/bench.py --steps 4 --warmup 8 --no-cpu-baseline --no-roofline > $R/gpurun_out/r2_trace.json 2> $R/gpurun_out/r2_trace.err
cd $R
ls -la gpurun_out/r2_trace/
python tools/timeline2.py gpurun_out/r2_trace/tr_kernel_trace.csv
