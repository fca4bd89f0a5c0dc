set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "backbone or gradients_fp64" > gpurun_out/r2_t6.log 2>&1 || { tail -40 gpurun_out/r2_t6.log; exit 1; }
tail -3 gpurun_out/r2_t6.log
python tools/exp_classes.py - MMNN_WG3_SPLIT_CAP=32 MMNN_WG3_SPLIT_CAP=48 - MMNN_WG3_SPLIT_CAP=32
