#!/usr/bin/env python3
"""Summarise a rocprofv3 output dir produced by scripts/gpu_pmc.sh into <dir>/summary.txt:
per-kernel stats (kernel-trace) and per-kernel mean of every PMC counter collected."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
out = []
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    out.append("== kernel stats (%s)" % os.path.relpath(f, d))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12:
            out.append("  " + ", ".join(row))
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out.append("== pmc (%s)" % os.path.relpath(f, d))
    for k, cs in agg.items():
        if "roi_align" in k or "iou" in k or "nms" in k or "deform" in k:
            for c, v in cs.items():
                out.append("  %-70s %-32s mean %.6g over %d dispatches" % (k, c, sum(v) / len(v), len(v)))
open(os.path.join(d, "summary.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
