#!/usr/bin/env python3
"""Per-kernel averages of every counter in rocprofv3 counter_collection.csv files (perf-debug helper):
    python tools/pmc_quick.py DIR [substring-of-kernel-name ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, pats = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if pats and not any(p in k for p in pats):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    for k in sorted(acc):
        print(k[:90], {c: round(v / max(1, len(disp[(k, c)])), 1) for c, v in sorted(acc[k].items())})


if __name__ == "__main__":
    main()
