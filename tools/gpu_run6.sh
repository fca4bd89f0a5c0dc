set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "backbone_forward_backward or tile_matrix or repeated or config3_backbone" > gpurun_out/r2_t8.log 2>&1 || { tail -40 gpurun_out/r2_t8.log; exit 1; }
tail -3 gpurun_out/r2_t8.log
python tools/exp_classes.py - -
python tools/phase_trace.py > gpurun_out/r2_phase_trace2.txt 2>&1
