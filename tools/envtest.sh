cd $GRAFT_REPO_ROOT
export PICK=conv2_fwd.b2,conv2_dgrad.b2,conv2_fwd.b1,conv1_fwd.b2,stem_conv
timeout -k 10 400 python -m pytest tests/test_backbone_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -3
for e in "X=1" "X=1"; do
  python tools/exp_classes.py "$e"
done
