"""Run bench.py once per experiment setting and print step time + the per-class kernel table (developer tool).
usage: python tools/exp_classes.py "ENV=VAL ENV2=VAL" "..."   (use "-" for the default environment)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    env = dict(os.environ)
    if spec != "-":
        for kv in spec.split():
            k, v = kv.split("=", 1)
            env[k] = v
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "8", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True)
    if r.returncode != 0:
        print(spec, "FAILED", r.stderr[-800:])
        continue
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cls = {c["class"]: c for c in d["roofline"]["classes"]}
    pick = os.environ.get("PICK", "conv2_fwd.b1,conv2_dgrad.b1,conv2_wgrad.b1,conv2_fwd.b2,conv2_dgrad.b2,conv1_dgrad.b1,stem_conv,stem_wgrad").split(",")
    if os.environ.get("ALL_CLASSES") == "1":
        print(f"{spec}: {d['ms_per_step']:.3f} ms/step, conv kernels {d['roofline']['conv_ms_per_step']:.3f} ms/step (single stream)")
        for c in d["roofline"]["classes"]:
            print(f"   {c['class']:16s} {c['ms_per_step']:.3f} ms  {c['launches'] // 4:3d}x {c['avg_us']:7.1f} us  {c['frac']:.3f} of peak")
        continue
    print(f"{spec:40s} {d['ms_per_step']:.3f} ms/step  " + "  ".join(f"{k}={cls[k]['avg_us']:.1f}us/{cls[k]['frac']:.2f}" for k in pick if k in cls), flush=True)
