#!/usr/bin/env python3
"""Steady-state per-step kernel breakdown from a rocprofv3 --kernel-trace CSV.

usage: steady_state.py <kernel_trace.csv> <marker substring> <marker launches per step> [steps] [rows]
The window is the last `steps` complete steps, delimited by every n-th launch of the marker kernel, so the
warm-up (MIOpen solver search, allocator growth) is excluded."""
import csv
import sys
from collections import defaultdict

path, marker, per_step = sys.argv[1], sys.argv[2], int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
top = int(sys.argv[5]) if len(sys.argv) > 5 else 40
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
bounds = marks[::per_step]
assert len(bounds) > steps, "not enough steps in the trace"
lo, hi = bounds[-steps - 1], bounds[-1]
win = rows[lo:hi]
t0, t1 = win[0][0], rows[hi][0]
agg = defaultdict(lambda: [0, 0])
for s, e, n in win:
    agg[n][0] += e - s
    agg[n][1] += 1
busy = sum(v[0] for v in agg.values())
print("window per step %.3f ms, kernel busy per step %.3f ms, launches per step %.1f" %
      ((t1 - t0) / 1e6 / steps, busy / 1e6 / steps, len(win) / steps))
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%6.2f%% %8.3f ms/step %6.1f calls/step avg %8.1f us  %s" %
          (100.0 * d / busy, d / 1e6 / steps, c / steps, d / c / 1e3, n[:120]))
