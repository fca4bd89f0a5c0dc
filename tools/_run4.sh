cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
rocprofv3 --kernel-trace -d $O/tr -o t --output-format csv -- python3 $R/bench.py --config gradcam256 --size 128 --steps 2 --warmup 2 > $O/gc.json 2> $O/gc.err
python3 - <<PY
import csv, glob, re
f = glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"mmnn::|void |\(.*", "", r["Kernel_Name"])[:60] for r in rows]
# last forward: find the last stem_conv
idx = max(i for i, n in enumerate(names) if n.startswith("stem_conv"))
out = open("$O/seq.txt", "w")
for i in range(max(0, idx - 5), min(len(names), idx + 200)):
    out.write(f"{i} {names[i]} grid={rows[i].get('Grid_Size','?')} wg={rows[i].get('Workgroup_Size','?')}\n")
out.close()
PY
head -80 $O/seq.txt
find $O/tr -name "*.csv" -delete
