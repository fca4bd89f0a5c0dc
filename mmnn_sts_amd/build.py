"""Build libmmnn_sts.so (hand-written HIP for gfx950) in-tree with hipcc.  `python -m mmnn_sts_amd.build`.

No torch.utils.cpp_extension (it hipifies), no cmake: one object per .hip file, compiled in parallel, linked into
mmnn_sts_amd/libmmnn_sts.so.  An object is rebuilt only when its source or a header it includes (transitively) is newer.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmmnn_sts.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp"))


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps_mtime(path, seen=None):
    """Newest modification time of `path` and of every project header it includes (transitively): a header edit rebuilds
    only the translation units that see it (the big convolution TUs take minutes each)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return 0.0
    seen.add(path)
    t = os.path.getmtime(path)
    for inc in _INC.findall(open(path).read()):
        t = max(t, _deps_mtime(os.path.join(os.path.dirname(path), inc), seen))
    return t


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    trace = os.environ.get("MMNN_PHASE_TRACE") == "1"     # developer build (tools/phase_trace.py): its own objects and library
    fenced = os.environ.get("MMNN_KZ_FENCED") == "1"      # portable K-split hand-off (csrc/fprop.hpp): its own objects and library
    flags = FLAGS + (["-DMMNN_PHASE_TRACE"] if trace else []) + (["-DMMNN_KZ_FENCED=1"] if fenced else [])
    variant = "_trace" if trace else ("_fenced" if fenced else "")
    OBJ = os.path.join(HERE, "build" + variant)
    LIB = os.path.join(HERE, f"libmmnn_sts{variant}.so")
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < _deps_mtime(s):
            lang = ["-x", "hip"] if src.endswith(".hip") else []
            jobs.append([hipcc, *flags, *lang, "-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    if jobs:
        if verbose:
            print(f"[mmnn_sts_amd.build] compiling {len(jobs)} file(s) for {ARCH}", flush=True)
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, len(jobs))) as ex:
            for warn in ex.map(run, jobs):
                if warn and verbose:
                    sys.stderr.write(warn)
    if jobs or not os.path.exists(LIB) or force:
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs])
        if verbose:
            print(f"[mmnn_sts_amd.build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
