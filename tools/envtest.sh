cd $GRAFT_REPO_ROOT
for e in "X=1" "AMD_DIRECT_DISPATCH=0" "X=1" "AMD_DIRECT_DISPATCH=0"; do
  echo "== $e"; env $e python tools/phase_times.py 2>&1 | tail -1
done
