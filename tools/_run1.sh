set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r3a/gpu_tests.log 2>&1 || { tail -60 gpurun_out/r3a/gpu_tests.log; exit 1; }
tail -15 gpurun_out/r3a/gpu_tests.log
python bench.py > gpurun_out/r3a/bench_line.json 2> gpurun_out/r3a/bench.err || { tail -30 gpurun_out/r3a/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3a/bench_line.json'))
print({k:v for k,v in d.items() if k not in('roofline','cpu_baseline','config')})
r=d['roofline']; print({k:v for k,v in r.items() if k not in ('classes','bounds','timing','kernel')})
for c in r['classes']: print(f"{c['class']:16s} {c['ms_per_step']:.3f} ms {c['launches']//4:3d}x {c['avg_us']:7.1f} us {c['bound']:5s} mfma {c['mfma_frac']:.3f} hbm {c['hbm_frac']:.3f} {c.get('limiter','')}")
print(d['cpu_baseline'])
PY
python tools/step_ops.py 64 3 > gpurun_out/r3a/step_ops.txt 2>&1 || tail -20 gpurun_out/r3a/step_ops.txt
head -60 gpurun_out/r3a/step_ops.txt
