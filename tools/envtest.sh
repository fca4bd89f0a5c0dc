cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "backbone_forward_backward or tile_matrix or repeated or side_streams" 2>&1 | tail -2
for e in "X=1" "MMNN_WGRAD_GROUP=1,4,8" "X=1" "MMNN_WGRAD_GROUP=2,4,8" "MMNN_WGRAD_GROUP=3,6,12" "MMNN_WGRAD_GROUP=1,0,0"; do
  echo "== $e"; env $e python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value'],1))"
done
PICK=conv2_wgrad.b1,conv2_wgrad.b2,conv2_wgrad.b3,conv1_wgrad.b1,conv1_wgrad.b2,conv1_wgrad.b3 python tools/exp_classes.py -
