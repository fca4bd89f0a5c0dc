set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "backbone_forward_backward or tile_matrix or repeated" > gpurun_out/r2_t10.log 2>&1 || { tail -40 gpurun_out/r2_t10.log; exit 1; }
tail -3 gpurun_out/r2_t10.log
python tools/exp_classes.py - -
python tools/phase_times.py 2>&1 | tail -1
