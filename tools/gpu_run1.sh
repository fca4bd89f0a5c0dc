set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x --durations=5 -k "two_ranks" > gpurun_out/r2_t2.log 2>&1 || { tail -40 gpurun_out/r2_t2.log; exit 1; }
tail -8 gpurun_out/r2_t2.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
cat gpurun_out/r2_bench1.json
MMNN_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 3 --size 64 > gpurun_out/r2_bench_g2.json 2> gpurun_out/r2_bench_g2.err || { tail -20 gpurun_out/r2_bench_g2.err; exit 1; }
cat gpurun_out/r2_bench_g2.json
cd /tmp && export TMPDIR=/tmp
MMNN_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2_prof_ss -o ss --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r2_prof_ss.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2_prof_ss.err
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2_prof_ov -o ov --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r2_prof_ov.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2_prof_ov.err
cd $GRAFT_REPO_ROOT
find gpurun_out/r2_prof_ss gpurun_out/r2_prof_ov -name "*kernel_trace.csv" -delete
find gpurun_out/r2_prof_ss gpurun_out/r2_prof_ov -type f | head
cat gpurun_out/r2_prof_ss.json
