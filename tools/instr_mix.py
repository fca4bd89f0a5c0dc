"""Instruction mix of the K loops of the convolution kernels (developer tool, CPU only): disassembles the gfx950 code objects of the
in-tree build and counts, between the first and the last MFMA of every fprop / wgrad kernel, the instructions by class.
    python tools/instr_mix.py [name-filter ...] > profiles/rNN_instruction_mix.txt"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
CXXFILT = "/usr/bin/c++filt"
objs = [f for f in sorted(os.listdir(os.path.join(ROOT, "mmnn_sts_amd", "build"))) if f.endswith(".o") and f.startswith(("fprop_inst_27", "wgrad_k3", "stem"))]
filters = sys.argv[1:]
tmp = tempfile.mkdtemp()
rows = []
try:
    for o in objs:
        shutil.copy(os.path.join(ROOT, "mmnn_sts_amd", "build", o), os.path.join(tmp, o))
        subprocess.run([OBJDUMP, "--offloading", os.path.join(tmp, o)], capture_output=True)
        cos = [f for f in os.listdir(tmp) if f.startswith(o + ".") and "gfx950" in f]
        if not cos:
            continue
        asm = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, cos[0])], capture_output=True, text=True).stdout
        for name, body in re.findall(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", asm, flags=re.S | re.M):
            dem = subprocess.run([CXXFILT, name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"mmnn::|void |\(.*", "", dem)
            if filters and not any(f in dem for f in filters):
                continue
            ins = [l.split("//")[0].split()[0] for l in body.splitlines() if l.strip() and not l.strip().startswith("//") and l.split("//")[0].split()]
            mf = [i for i, x in enumerate(ins) if x.startswith("v_mfma")]
            if len(mf) < 8:
                continue
            seg = ins[mf[0]:mf[-1] + 1]
            cnt = {"mfma": 0, "valu": 0, "salu": 0, "ds_read": 0, "ds_write": 0, "vmem_load": 0, "vmem_store": 0, "waitcnt": 0, "barrier": 0, "other": 0}
            for x in seg:
                if x.startswith("v_mfma"): cnt["mfma"] += 1
                elif x.startswith("v_"): cnt["valu"] += 1
                elif x.startswith("s_waitcnt"): cnt["waitcnt"] += 1
                elif x.startswith("s_barrier"): cnt["barrier"] += 1
                elif x.startswith("s_"): cnt["salu"] += 1
                elif x.startswith(("ds_read", "ds_load", "ds_swizzle", "ds_bpermute")): cnt["ds_read"] += 1
                elif x.startswith("ds_"): cnt["ds_write"] += 1
                elif x.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")): cnt["vmem_load"] += 1
                elif x.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")): cnt["vmem_store"] += 1
                else: cnt["other"] += 1
            rows.append((dem[:78], cnt))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
print("instructions between the first and the last MFMA of each kernel (static count over the unrolled loop bodies, every path once)")
print(f"{'kernel':78s} {'mfma':>5s} {'valu':>5s} {'valu/mfma':>9s} {'salu':>5s} {'ds_rd':>5s} {'ds_wr':>5s} {'vld':>4s} {'vst':>4s} {'wait':>5s} {'bar':>4s}")
for name, c in sorted(rows):
    print(f"{name:78s} {c['mfma']:5d} {c['valu']:5d} {c['valu'] / c['mfma']:9.2f} {c['salu']:5d} {c['ds_read']:5d} {c['ds_write']:5d} {c['vmem_load']:4d} {c['vmem_store']:4d} "
          f"{c['waitcnt']:5d} {c['barrier']:4d}")
