#!/usr/bin/env python3
"""HBM bytes per launch per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/<round>_traffic.json (round tag: $EGOEGO_ROUND, default r03).

usage: traffic_json.py <counter_collection.csv of the FETCH_SIZE pass> <... of the WRITE_SIZE pass> B T precision [suffix]
(suffix, e.g. "_p3": the file is profiles/<round>_traffic<suffix>.json and is marked weights = "any" — HBM bytes of a precision do not depend on the weights)
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests of wide coalesced reads are counted as 64 B)."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(path, name):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == name:
            acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, B, T, prec = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    suffix = sys.argv[6] if len(sys.argv) > 6 else ""
    fetch, write = mean_counter(fetch_csv, "FETCH_SIZE"), mean_counter(write_csv, "WRITE_SIZE")
    out = {"_provenance": "rocprofv3 --kernel-trace --pmc (separate passes: 'GRBM_GUI_ACTIVE FETCH_SIZE', 'WRITE_SIZE') -- python3 bench.py "
                          "--steps 3 --warmup 1 --precision N --no-probe --no-cpu-baseline --no-graph; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled per "
                          "MI355X_MICROARCH.md (gfx950 counts 128-B requests of wide coalesced reads as 64 B); written by tools/traffic_json.py",
           "batch": B, "window": T, "precision": prec, "kernels": {}}
    if suffix:
        out["weights"] = "any"
    for k in sorted(set(fetch) | set(write)):
        if k.startswith("void at::") or k.startswith("__amd"):  # torch / runtime helpers of bench.py's setup, not the measured path
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out["kernels"][k] = {"fetch_kb_raw": f, "write_kb": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
    with open(os.path.join(ROOT, "profiles", os.environ.get("EGOEGO_ROUND", "r06") + f"_traffic{suffix}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:45s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
