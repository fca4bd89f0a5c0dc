#!/bin/bash
# Evidence run on the GPU box (developer tool):  gpurun --timeout 1200 -- 'bash tools/gpu_final.sh r03 [tests|bench|prof|pmc ...]'
# Stages (default: all): tests = full GPU suite; bench = the bench line; prof = rocprofv3 kernel statistics of the same command;
# pmc = the counter passes (separate passes per counter group, kernel trace only).  Everything lands under
# gpurun_out/<tag>_final/ ; tools/collect_profiles.py <dir> <tag> condenses it into profiles/.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
TAG="${1:-r03}"; shift || true
STAGES="${*:-tests bench prof pmc}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/${TAG}_final"
mkdir -p "$O"
cd "$R"
has() { case " $STAGES " in *" $1 "*) return 0;; *) return 1;; esac; }
if has tests; then
  python -m pytest tests -m gpu -q --durations=8 > "$O/gpu_tests.log" 2>&1 || { tail -40 "$O/gpu_tests.log"; exit 1; }
  tail -12 "$O/gpu_tests.log"
fi
if has bench; then
  python bench.py > "$O/bench_line.json" 2> "$O/bench_line.err"
  python tools/show_bench.py "$O/bench_line.json"
  python bench.py --config unimodal --no-cpu-baseline > "$O/bench_unimodal.json" 2> "$O/bench_unimodal.err"
  python tools/show_bench.py "$O/bench_unimodal.json" > "$O/bench_unimodal.txt"
  python bench.py --config gradcam256 --steps 10 --warmup 3 > "$O/bench_gradcam256.json" 2> "$O/bench_gradcam256.err"
  cat "$O/bench_gradcam256.json"
  ./tools/microbench/grid_barrier.bin > "$O/grid_barrier.txt" 2>&1 || true
  ./tools/microbench/mfma_valu_overlap.bin > "$O/mfma_valu_overlap.txt" 2>&1 || true
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  rocprofv3 --kernel-trace --stats -d "$O/prof_ss" -o ss --output-format csv -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/prof_ss.err"
  rocprofv3 --kernel-trace --stats -d "$O/prof_gc" -o gc --output-format csv -- python3 "$R/bench.py" --config gradcam256 --steps 5 --warmup 2 > "$O/bench_gradcam256_under_rocprof.json" 2> "$O/prof_gc.err"
  find "$O/prof_gc" -name "*kernel_trace.csv" -delete
  find "$O/prof_ss" -name "*kernel_trace.csv" -delete
fi
pmc_pass() { local name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" -d "$O/pmc_$name" -o p --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$O/pmc_$name.log" 2>&1 || echo "pass $name failed"
  find "$O/pmc_$name" -name "*kernel_trace.csv" -delete
}
if has pmc; then
  pmc_pass fetch FETCH_SIZE
  pmc_pass write WRITE_SIZE
  pmc_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
  pmc_pass sq2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM
fi
du -sh "$O"
