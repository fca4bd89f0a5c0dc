cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a
mkdir -p $O
T="tests/test_backbone_gpu.py::test_backbone_forward_backward"
for e in "MMNN_WG3_NO_XCD=1 MMNN_WG1_NO_XCD=1" "MMNN_WG3_NO_XCD=1" "MMNN_WG1_NO_XCD=1" "X=1"; do
  echo "=== $e"
  env $e AMD_LOG_LEVEL=1 timeout -k 10 300 python -m pytest "$T" -x -q -k "blocks0" > $O/bisect.log 2>&1
  echo "rc=$?"; grep -v "^  File\|^Extension" $O/bisect.log | tail -8
done
