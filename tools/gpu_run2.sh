set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "backbone or gradients_fp64 or two_ranks" > gpurun_out/r2_t4.log 2>&1 || { tail -40 gpurun_out/r2_t4.log; exit 1; }
tail -3 gpurun_out/r2_t4.log
for cfg in "nobatch:1,2,4" "1,4,8" "1,6,12" "1,12,24" "2,12,24" "1,2,4"; do
  g=${cfg#nobatch:}
  if [ "$g" != "$cfg" ]; then export MMNN_NO_WGRAD_BATCH=1; else unset MMNN_NO_WGRAD_BATCH; fi
  echo "== $cfg"
  MMNN_WGRAD_GROUP=$g python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
