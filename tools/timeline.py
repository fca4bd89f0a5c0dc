"""Per-queue timeline of the last backward in a rocprofv3 kernel_trace.csv (developer tool)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nm = lambda r: re.sub(r'mmnn::|void |\(.*', '', r['Kernel_Name'])[:58]
# last backward = from the last consumer_bwd mode-0 (first kernel after the last bn_apply) to the last finalize
ia = max(i for i, r in enumerate(rows) if nm(r).startswith('bn_apply'))
ib = max(i for i, r in enumerate(rows) if nm(r).startswith('finalize'))
seg = rows[ia + 1: ib + 1]
t0 = int(seg[0]['Start_Timestamp'])
qs = sorted({r['Queue_Id'] for r in seg})
print("queues", qs, "span %.3f ms" % ((int(seg[-1]['End_Timestamp']) - t0) / 1e6))
for q in qs:
    rs = [r for r in seg if r['Queue_Id'] == q]
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.3f} ms, first {(int(rs[0]['Start_Timestamp']) - t0) / 1e3:.0f} us, last end {(int(rs[-1]['End_Timestamp']) - t0) / 1e3:.0f} us")
if len(sys.argv) > 2:
    lo, hi = float(sys.argv[2]), float(sys.argv[3])
    for r in seg:
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        if e >= lo and s <= hi:
            print(f"q{r['Queue_Id']} {s:9.1f} {e:9.1f} {e - s:7.1f}  {nm(r)}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
