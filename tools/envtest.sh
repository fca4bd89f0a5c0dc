cd $GRAFT_REPO_ROOT
export ALL_CLASSES=1
for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "X=1"; do
  echo "== $e"; python tools/exp_classes.py "$e" 2>&1
done
