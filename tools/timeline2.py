"""Whole-step timeline from a rocprofv3 kernel_trace.csv (developer tool): python tools/timeline2.py <csv> [dump_lo_us dump_hi_us]
Takes the last complete step (stem_conv .. sgd_kernel), prints per-queue busy time, the main queue's idle gaps by phase and the
largest gaps."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nm = lambda r: re.sub(r'mmnn::|void |\(.*', '', r['Kernel_Name'])[:64]
stems = [i for i, r in enumerate(rows) if nm(r).startswith('stem_conv')]
sgds = [i for i, r in enumerate(rows) if nm(r).startswith('sgd_kernel')]
ib = sgds[-1]
ia = max(i for i in stems if i < ib)
ia = max(i for i in range(ia) if nm(rows[i]).startswith('pack_kernel')) if any(nm(rows[i]).startswith('pack_kernel') for i in range(ia)) else ia
seg = rows[ia: ib + 1]
t0 = int(seg[0]['Start_Timestamp'])
S = lambda r: (int(r['Start_Timestamp']) - t0) / 1e3
E = lambda r: (int(r['End_Timestamp']) - t0) / 1e3
print("step span %.3f ms, %d kernels" % (E(seg[-1]) / 1e3, len(seg)))
qs = sorted({r['Queue_Id'] for r in seg})
mainq = max(qs, key=lambda q: sum(1 for r in seg if r['Queue_Id'] == q))
for q in qs:
    rs = [r for r in seg if r['Queue_Id'] == q]
    print(f"queue {q}{' (main)' if q == mainq else ''}: {len(rs)} kernels, busy {sum(E(r) - S(r) for r in rs) / 1e3:.3f} ms, first {S(rs[0]):.0f} us, last end {E(rs[-1]):.0f} us")
main = [r for r in seg if r['Queue_Id'] == mainq]
# phases on the main queue
def phase_of(i, r):
    return None
bn_apply = next((i for i, r in enumerate(main) if nm(r).startswith('bn_apply')), None)
cons = [i for i, r in enumerate(main) if nm(r).startswith('consumer_bwd')]
marks = [("forward", 0, bn_apply)]
names = ["block4 bwd", "block3 bwd", "block2 bwd", "block1 bwd"]
# backward starts at first consumer_bwd after bn_apply
cb = [i for i in cons if i > bn_apply]
spb = next((i for i, r in enumerate(main) if nm(r).startswith('stem_pool_bwd')), len(main) - 1)
bounds = cb + [spb]
marks.append(("tail fwd+loss", bn_apply + 1, cb[0] - 1))
for k in range(len(cb)):
    marks.append((names[k] if k < 4 else f"phase{k}", cb[k], bounds[k + 1] - 1))
marks.append(("stem bwd + finalize + sgd", spb, len(main) - 1))
for name, a, b in marks:
    rs = main[a: b + 1]
    if not rs:
        continue
    span = E(rs[-1]) - S(rs[0])
    busy = sum(E(r) - S(r) for r in rs)
    print(f"{name:28s} {S(rs[0]):9.0f} -> {E(rs[-1]):9.0f} us  span {span:8.0f}  main-queue busy {busy:8.0f}  idle {span - busy:7.0f}  kernels {len(rs)}")
gaps = sorted(((S(main[i + 1]) - E(main[i]), i) for i in range(len(main) - 1)), reverse=True)[:12]
print("largest main-queue gaps (us): ", [(round(g, 1), nm(main[i])[:28], '->', nm(main[i + 1])[:28]) for g, i in gaps])
if len(sys.argv) > 3:
    lo, hi = float(sys.argv[2]), float(sys.argv[3])
    for r in seg:
        if E(r) >= lo and S(r) <= hi:
            print(f"q{r['Queue_Id']} {S(r):9.1f} {E(r):9.1f} {E(r) - S(r):7.1f}  {nm(r)}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
