"""Summarise a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> <steps>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot / 1e6 / steps:.3f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 32]:
    name = re.sub(r'mmnn::|void |\(.*', '', r['Name'])[:70]
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms {int(r['Calls']) / steps:6.1f}x {float(r['AverageNs']) / 1e3:8.1f} us {float(r['Percentage']):5.1f}%  {name}")
