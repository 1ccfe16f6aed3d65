#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"void (\w+)<", name)
    base = m.group(1) if m else name.split("(")[0]
    for tag in ("EpiEmbed", "EpiOut", "EpiQK") + (("EpiTiled", "EpiResLN") if "layer_tail" not in name else ()):  # (layer_tail*: one tag per kernel)
        if tag in name:
            base += ":" + tag
    return base


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        for row in csv.DictReader(open(p)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    counters = sorted({c for k in acc.values() for c in k})
    print("kernel," + ",".join(counters) + ",launches")
    for k, d in sorted(acc.items()):
        if k.startswith("void at::") or k.startswith("__amd"):  # torch / runtime helpers of bench.py's setup
            continue
        n = max(len(v) for v in d.values())
        print(k + "," + ",".join(f"{sum(d[c]) / len(d[c]):.4g}" if c in d else "" for c in counters) + f",{n}")


if __name__ == "__main__":
    main(sys.argv[1:])
