#!/usr/bin/env python3
"""Per-iteration summary of a rocprofv3 --kernel-trace --stats CSV: tools/kstats.py <kernel_stats.csv> <iterations>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
it = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(int(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"total {tot / it / 1e6:.2f} ms/iter, {calls / it:.0f} launches/iter")
aten = [r for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"]]
print(f"ATen/copy kernels: {sum(int(r['Calls']) for r in aten) / it:.0f} launches, "
      f"{sum(int(r['TotalDurationNs']) for r in aten) / it / 1e6:.2f} ms per iter")
for r in rows[:top]:
    name = r["Name"].replace("siv::", "").replace("void ", "")
    name = name[:name.index("(")] if "(" in name else name
    print(f"{name[:58]:58s} {int(r['Calls']) / it:7.1f} {int(r['TotalDurationNs']) / it / 1e6:8.2f} ms  avg {float(r['AverageNs']) / 1e3:8.1f} us")
