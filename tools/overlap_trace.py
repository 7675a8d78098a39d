#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints, for the k_run256v2 dispatches, start / end relative to the previous
dispatch: shows whether consecutive launches of the pipelined entry point overlap."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if "k_run256v2" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
rows.sort()
print(len(rows), "dispatches")
prev = None
for i, (s, e, q) in enumerate(rows):
    if prev and (i < 12 or i > len(rows) - 40):
        print(f"{i:4d} queue {q}: start {(s - prev[0]) / 1e3:8.1f} us after the previous start, {(s - prev[1]) / 1e3:8.1f} us after its end; duration {(e - s) / 1e3:7.1f} us")
    prev = (s, e)
