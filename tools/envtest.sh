cd $GRAFT_REPO_ROOT
export PICK=conv2_wgrad.b1,conv2_wgrad.b2,conv2_wgrad.b3,stem_wgrad
for e in "X=1" "MMNN_WG3_SPLIT_CAP=10" "MMNN_WG3_SPLIT_CAP=16" "MMNN_WG3_SPLIT_CAP=128" "X=1"; do
  python tools/exp_classes.py "$e"
done
