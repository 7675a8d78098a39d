#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output under a profile directory (tools/profile.sh):
kernel stats + per-kernel mean of every PMC counter.  Usage: tools/pmc_summary.py DIR"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace("csdr::(anonymous namespace)::", "").replace("csdr::", "").replace("void ", "")
    return n.split("(")[0][:60]


def main(d):
    for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        print(f"## kernel stats ({os.path.relpath(f, d)})")
        for row in csv.DictReader(open(f)):
            nm = short(row.get("Name", ""))
            if any(s in nm for s in ("k_", "Cijk", "rocclr")):
                print(f"  {nm:45s} calls={row.get('Calls'):>5s} total_ns={row.get('TotalDurationNs'):>12s} "
                      f"avg_ns={float(row.get('AverageNs', 0)):>12.0f} min={row.get('MinNs')} max={row.get('MaxNs')} pct={row.get('Percentage')}")
    for sub in sorted(glob.glob(os.path.join(d, "pmc*"))):
        if not os.path.isdir(sub):
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if acc:
            print(f"## {os.path.basename(sub)}: mean counter value per dispatch")
        for k, cs in acc.items():
            print(f"  {k}")
            for c, v in sorted(cs.items()):
                print(f"      {c:28s} {sum(v) / len(v):18.1f}   (n={len(v)})")


if __name__ == "__main__":
    main(sys.argv[1])
