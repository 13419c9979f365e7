#!/usr/bin/env python3
"""Concurrency of kernels over time from a rocprofv3 kernel trace (kernel_trace.csv):
how long the GPU ran nothing, one kernel, two kernels ..., split by which kernels
(T = traversal, S = shading, o = other), over the busiest contiguous part of the run.

  python tools/timeline.py <kernel_trace.csv> [skip_fraction]
"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = []
with open(path) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"]
        kind = "T" if "k_trace" in name else "S" if "k_shade" in name else "o"
        ev.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), kind))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + (t1 - t0) * skip  # skip set-up, counting pass and warm-up
points = []
for s, e, k in ev:
    if e <= lo:
        continue
    points.append((max(s, lo), 1, k))
    points.append((e, -1, k))
points.sort()
active = defaultdict(int)
hist = defaultdict(float)
last = lo
for t, d, k in points:
    key = "".join(sorted(kk * active[kk] for kk in active if active[kk])) or "-"
    hist[key] += t - last
    last = t
    active[k] += d
total = sum(hist.values())
print(f"window {total / 1e9:.3f} s")
for key, v in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(f"  {key:8s} {v / 1e6:10.1f} ms  {100 * v / total:5.1f} %")
dur = defaultdict(list)
for s, e, k in ev:
    if s >= lo:
        dur[k].append(e - s)
for k, v in dur.items():
    print(f"  {k}: {len(v)} launches, mean {sum(v) / len(v) / 1e6:.3f} ms, total {sum(v) / 1e9:.3f} s")
