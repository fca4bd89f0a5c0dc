set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a
mkdir -p $O
python -m pytest tests/test_kz_handoff_gpu.py -q -s > $O/kz_tests.log 2>&1 || { tail -40 $O/kz_tests.log; exit 1; }
tail -8 $O/kz_tests.log
python bench.py > $O/bench_line.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python tools/show_bench.py $O/bench_line.json
./tools/microbench/grid_barrier.bin > $O/grid_barrier.txt 2>&1 || true
cat $O/grid_barrier.txt
./tools/microbench/mfma_valu_overlap.bin > $O/mfma_valu_overlap.txt 2>&1 || true
head -40 $O/mfma_valu_overlap.txt
PICK=conv2_dgrad.b1,conv2_wgrad.b1,conv1_wgrad.b1,conv1_wgrad.b2,conv2_wgrad.b2 python tools/exp_classes.py "-" "MMNN_DGRAD_TILE=1" "MMNN_WG3_NO_XCD=1 MMNN_WG1_NO_XCD=1" > $O/ab.txt 2>&1 || true
cat $O/ab.txt
python tools/step_ops.py 64 3 > $O/step_ops.txt 2>&1 || tail -20 $O/step_ops.txt
head -50 $O/step_ops.txt
