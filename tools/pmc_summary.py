#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + PMC passes) for the step kernel."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
kern = sys.argv[2] if len(sys.argv) > 2 else "drv_step_kernel"
for f in sorted(glob.glob(root + "/*/runc/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        n = len(v)
        print("%-24s n=%d mean=%.4g first50=%.4g last50=%.4g" % (c, n, sum(v) / n, sum(v[:50]) / 50, sum(v[-50:]) / 50))
for f in sorted(glob.glob(root + "/kt/runc/*kernel_trace.csv")):
    d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
         for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    d.sort()
    dur = [x[1] for x in d]
    print("dispatch durations (us) by step index:")
    for i in range(0, len(dur), 50):
        seg = dur[i:i + 50]
        print("  steps %4d-%4d: mean %.1f" % (i, i + len(seg) - 1, sum(seg) / len(seg) / 1e3))
