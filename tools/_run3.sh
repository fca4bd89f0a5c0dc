set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_backbone_gpu.py -q -x -k "persistent or config3_backbone_128" > $O/persist_tests.log 2>&1 || { tail -60 $O/persist_tests.log; exit 1; }
tail -3 $O/persist_tests.log
PICK=block_fwd.b3,block_fwd.b4,conv1_fwd.b3,conv2_fwd.b3,conv1_fwd.b4,conv2_fwd.b4 python tools/exp_classes.py "-" "MMNN_PERSISTENT=1" > $O/ab_persist.txt 2>&1 || true
cat $O/ab_persist.txt
