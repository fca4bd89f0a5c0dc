cd $GRAFT_REPO_ROOT
for e in "MMNN_WG1_BLOCKS=1536" "MMNN_WG1_BLOCKS=2048" "MMNN_WG1_BLOCKS=3072" "MMNN_WG1_BLOCKS=4096"; do
  echo "== $e"; env $e PICK=conv1_wgrad.b1,conv1_wgrad.b2,conv1_wgrad.b3,conv1_wgrad.b4 python tools/exp_classes.py - | cut -c40-
done
