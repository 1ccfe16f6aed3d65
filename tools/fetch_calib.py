#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE (KiB, rocprofv3 --pmc) of tools/microbench/fetch_calib.hip's three known-byte kernels -> the factor to
multiply each counter with for this library's access patterns.   usage: fetch_calib.py <FETCH pass csv> <WRITE pass csv>"""
import csv
import json
import sys
from collections import defaultdict

KNOWN = 256 * (8 << 20)


def means(path, counter):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter and row["Kernel_Name"].startswith("calib_"):
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


f, w = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
out = {"known_bytes_per_launch": KNOWN,
       "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
       "true_bytes_over_FETCH_SIZE_bytes": {k: KNOWN / (v * 1024) for k, v in f.items() if k != "calib_store" and v > 0},
       "true_bytes_over_WRITE_SIZE_bytes": {k: KNOWN / (v * 1024) for k, v in w.items() if k == "calib_store" and v > 0}}
print(json.dumps(out, indent=1))
