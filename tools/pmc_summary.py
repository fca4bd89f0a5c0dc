"""Per-kernel means of rocprofv3 --pmc counters (developer tool): python tools/pmc_summary.py <counter_collection.csv> [...more csv]
Prints one row per kernel name: launches and the mean of every counter found (summed over dimensions within a dispatch)."""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))    # kernel -> counter -> sum over dispatches
cnt = defaultdict(lambda: defaultdict(set))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = re.sub(r'mmnn::|void |\(.*', '', r['Kernel_Name'])[:72]
        c = r['Counter_Name']
        acc[k][c] += float(r['Counter_Value'])
        cnt[k][c].add((path, r['Dispatch_Id']))
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(72), "launches", *[n[:22].rjust(22) for n in names])
rows = []
for k in acc:
    n = max(len(cnt[k][c]) for c in cnt[k])
    rows.append((k, n, [acc[k][c] / max(1, len(cnt[k][c])) if c in acc[k] else float('nan') for c in names]))
rows.sort(key=lambda r: -r[1] * (r[2][0] if r[2] and r[2][0] == r[2][0] else 0))
for k, n, vals in rows[: int(__import__('os').environ.get('TOP', '40'))]:
    print(k.ljust(72), str(n).rjust(8), *[f"{v:22.1f}" for v in vals])
