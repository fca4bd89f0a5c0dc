cd $GRAFT_REPO_ROOT
for sz in 64 96 128; do for e in "MMNN_SIDE_STREAMS=2" "MMNN_SIDE_STREAMS=1" "MMNN_SIDE_STREAMS=0" "MMNN_SIDE_STREAMS=0 MMNN_WGRAD_GROUP=1,6,12" "MMNN_SIDE_STREAMS=1 MMNN_WGRAD_GROUP=1,6,12"; do
  echo "== size $sz $e"; env $e python bench.py --size $sz --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value'],1))"
done; done
