cd $GRAFT_REPO_ROOT
for e in "X=1" "MMNN_SIDE_STREAMS=1 MMNN_SIDE_FROM_BLOCK=2" "MMNN_SIDE_STREAMS=1 MMNN_SIDE_FROM_BLOCK=1" "X=1" "MMNN_SIDE_STREAMS=1 MMNN_SIDE_FROM_BLOCK=2" "MMNN_SIDE_STREAMS=1 MMNN_SIDE_FROM_BLOCK=3"; do
  echo "== $e"; env $e python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
