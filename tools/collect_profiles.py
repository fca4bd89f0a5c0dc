"""Condense gpurun_out/<tag>_final (tools/gpu_final.sh) into the tracked evidence under profiles/ (developer tool).
   python tools/collect_profiles.py [src_dir] [round_tag]"""
import csv, glob, io, json, os, re, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "r03_final")
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


for name, pat in (("bench_line.json", "bench_line.json"), ("bench_line_under_rocprof.json", "bench_under_rocprof.json"),
                  ("kernel_stats.csv", "prof_ss/**/*kernel_stats.csv"), ("gpu_tests.log", "gpu_tests.log"),
                  ("bench_line_unimodal.json", "bench_unimodal.json"), ("bench_line_gradcam256.json", "bench_gradcam256.json"),
                  ("kernel_stats_gradcam256.csv", "prof_gc/**/*kernel_stats.csv"), ("microbench_grid_barrier.txt", "grid_barrier.txt"),
                  ("microbench_mfma_valu_overlap.txt", "mfma_valu_overlap.txt"), ("ab_experiments.txt", "ab.txt"),
                  ("step_ops.txt", "step_ops.txt")):
    f = find(pat)
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_{name}"))

# per-kernel counter means
txt = io.StringIO()
for grp in ("fetch", "write", "sq1", "sq2"):
    f = find(f"pmc_{grp}/**/*counter_collection.csv")
    if not f:
        continue
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), f], capture_output=True, text=True, env=dict(os.environ, TOP="36")).stdout
    txt.write(f"# pass '{grp}': rocprofv3 --kernel-trace --pmc <counters below> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline\n")
    txt.write(f"# per-launch means (FETCH_SIZE / WRITE_SIZE in KB; SQ_* summed over the chip)\n{out}\n")
if txt.getvalue():
    open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w").write(txt.getvalue())


def mean_counter(grp, kernel_prefix, counter):
    f = find(f"pmc_{grp}/**/*counter_collection.csv")
    if not f:
        return None
    tot, ids = 0.0, set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r'mmnn::|void |\(.*', '', r['Kernel_Name'])
        if k.startswith(kernel_prefix) and r['Counter_Name'] == counter:
            tot += float(r['Counter_Value']); ids.add(r['Dispatch_Id'])
    return tot / len(ids) if ids else None


import bench  # noqa: E402  (algorithmic bytes per class: bench.class_bytes)
N, S = 2, 128
# kernel name prefix (as rocprofv3 prints it), class key, launches of the class per step, reads 16 B per lane?  (the gfx950 FETCH_SIZE
# correction -- a wide coalesced read is tallied at half its bytes, MI355X_MICROARCH.md HBM section -- applies to wide reads only)
kernels = {
    "conv2_fwd.b1": ("fprop_kernel<27, 1, 1, 1, 4, 1, 1, 2, 8, 2, 4, 32, true>", (1, 0), 6, True),
    "conv2_dgrad.b1": ("fprop_kernel<27, 2, 2, 2, 4, 1, 2, 2, 2, 2, 4, 32, false>", (2, 0), 6, True),
    "conv2_wgrad.b1": ("wgrad3_batched_kernel<1, 1, 2, 32>", (3, 0), 1, True),
    "conv2_wgrad.b2": ("wgrad3_batched_kernel<1, 1, 4, 16>", (3, 1), 1, True),
    "stem_conv": ("stem_conv_kernel<2>", (7, 0), 1, False),
    "stem_wgrad": ("stem_wgrad_kernel", (8, 0), 1, False),
    "sgd": ("sgd_kernel", None, 1, True),
    "pack": ("pack_kernel", None, 1, False),
}
res = {"correction": "hbm_bytes = (2 if wide_reads else 1) * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: gfx950 tallies a 16-byte-per-lane coalesced read at "
                     "half its bytes (MI355X_MICROARCH.md, HBM section); kernels that stage with 4-byte loads are reported uncorrected (uncalibrated)",
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline (separate passes)",
       "kernels": {}}
for cls, (kname, key, per_step, wide) in kernels.items():
    fe, wr = mean_counter("fetch", kname, "FETCH_SIZE"), mean_counter("write", kname, "WRITE_SIZE")
    if fe is None or wr is None:
        continue
    alg = bench.class_bytes(key[0], key[1], N, S) / per_step if key else None
    if cls == "sgd":
        alg = 4.0 * 11276902 * 5          # read p, g, momentum; write p, momentum
    hbm = (2 if wide else 1) * fe * 1024 + wr * 1024
    res["kernels"][cls] = {"kernel": kname, "FETCH_SIZE_KB_per_launch": fe, "WRITE_SIZE_KB_per_launch": wr, "wide_reads": wide,
                           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": (hbm / alg) if alg else None}
if res["kernels"]:
    json.dump(res, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, f"{tag}_traffic.json")).read())
