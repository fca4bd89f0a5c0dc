"""Condense gpurun_out/r2_final (tools/gpu_final.sh) into the tracked evidence under profiles/ (developer tool).
   python tools/collect_profiles.py [src_dir] [round_tag]"""
import csv, glob, io, json, os, re, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "r2_final")
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


for name, pat in (("bench_line.json", "bench_line.json"), ("bench_line_under_rocprof.json", "bench_under_rocprof.json"),
                  ("kernel_stats_two_side_streams.csv", "prof_ov/**/*kernel_stats.csv"), ("kernel_stats.csv", "prof_ss/**/*kernel_stats.csv"),
                  ("bench_line_two_side_streams.json", "bench_two_side_streams.json"),
                  ("gpu_tests.log", "gpu_tests.log")):
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


MB = 1e6
V1, N = 32 ** 3, 2                     # block-1 voxels per sample, micro-batch
t1 = N * 128 * V1 * 4                  # bottleneck tensor T1 (128 channels)
sl = N * 32 * V1 * 4                   # one 32-channel slice of the concat buffer
w2 = 27 * 32 * 128 * 4
kernels = {
    "conv2_fwd.b1": ("fprop_kernel<27, 1, 1, 1, 4, 1, 1, 2, 8, 2, 4, 32, true>", t1 + sl + w2, "read T1, write the 32 new channels, weights"),
    "conv2_dgrad.b1": ("fprop_kernel<27, 2, 2, 2, 2, 1, 2, 2, 2, 1, 4, 32, false>", 2 * sl + 2 * t1 + w2, "read G and X slices, read T1 (mask), write dZ2, weights"),
    "conv2_wgrad.b1": ("wgrad3_batched_kernel<1, 1, 2, 32>", 6 * (2 * sl + t1 + 21 * w2), "ONE launch for the 6 layers of block 1: per layer read G and X slices, read T1, write 21 partial slabs"),
}
res = {"correction": "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE counts wide reads at half size, MI355X_MICROARCH.md HBM section)",
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline (separate passes)",
       "kernels": {}}
for cls, (kname, alg, what) in kernels.items():
    fe, wr = mean_counter("fetch", kname, "FETCH_SIZE"), mean_counter("write", kname, "WRITE_SIZE")
    if fe is None or wr is None:
        continue
    res["kernels"][cls] = {"kernel": kname, "FETCH_SIZE_KB_per_launch": fe, "WRITE_SIZE_KB_per_launch": wr,
                           "hbm_bytes_per_launch": 2 * fe * 1024 + wr * 1024, "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_are": what}
json.dump(res, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_traffic.json")).read())
