# A/B harness for environment knobs on the GPU box (developer tool):  gpurun -- 'bash tools/envtest.sh > gpurun_out/envtest.log 2>&1'
# Each setting runs bench.py once and prints the step time plus the kernel classes named in PICK.
cd $GRAFT_REPO_ROOT
export PICK=${PICK:-conv2_fwd.b1,conv2_dgrad.b1,conv2_wgrad.b1,stem_conv,stem_wgrad}
for e in "X=1" "MMNN_SIDE_STREAMS=1" "X=1"; do
  python tools/exp_classes.py "$e"
done
