cd $GRAFT_REPO_ROOT
for e in "X=1" "MMNN_WG1_WC8_FROM=129" "MMNN_WG1_WC8_FROM=129 MMNN_WG1_BLOCKS=768" "MMNN_WG1_WC8_FROM=129 MMNN_WG1_BLOCKS=1024" "MMNN_WG1_WC8_FROM=1000"; do
  echo "== $e"; env $e PICK=conv1_wgrad.b1,conv1_wgrad.b2,conv1_wgrad.b3,conv1_wgrad.b4 python tools/exp_classes.py - | cut -c40-
done
