cd $GRAFT_REPO_ROOT
for e in "X=1" "MMNN_EXP_TILE1=65" "MMNN_EXP_TILE1=64"; do
  echo "== $e"; env $e PICK=conv1_fwd.b1,conv1_dgrad.b1,conv1_fwd.b2,conv1_dgrad.b2 python tools/exp_classes.py - | cut -c40-
done
