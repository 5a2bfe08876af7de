#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: wall time covered by at least one kernel, by two or more, and the
sum of kernel durations per stream (queue) -- how much of the side-stream work ran beside the main stream."""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
iters = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ev = []
per_q = collections.defaultdict(float)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
    per_q[r.get("Queue_Id", "?")] += (e - s)
ev.sort()
depth, last, cover, multi = 0, None, 0, 0
for t, d in ev:
    if depth >= 1:
        cover += t - last
    if depth >= 2:
        multi += t - last
    depth += d
    last = t
tot = sum(per_q.values())
print(f"sum of kernel durations {tot / 1e6 / iters:.2f} ms/iter, covered wall time {cover / 1e6 / iters:.2f} ms/iter, "
      f"two or more kernels at once {multi / 1e6 / iters:.2f} ms/iter")
for q, v in sorted(per_q.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: {v / 1e6 / iters:.2f} ms/iter")
