"""Print a bench.py JSON line as a table (developer tool): python tools/show_bench.py <file>"""
import json
import sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print({k: v for k, v in d.items() if k not in ("roofline", "cpu_baseline", "config")})
r = d.get("roofline")
if r:
    print({k: v for k, v in r.items() if k not in ("classes", "bounds", "timing", "kernel")})
    for c in r.get("classes", []):
        print(f"{c['class']:16s} {c['ms_per_step']:.3f} ms {c['launches'] // 4:3d}x {c['avg_us']:7.1f} us {c['bound']:5s} mfma {c['mfma_frac']:.3f} "
              f"hbm {c['hbm_frac']:.3f} {c.get('limiter', '')}")
print(d.get("cpu_baseline"))
