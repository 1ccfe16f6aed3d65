#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel, the mean kernel duration OF EACH PMC PASS
(from the pass's own Start/End timestamps: a profiled pass runs at a different clock than an un-profiled one, so rates must be
formed inside one pass), and what follows from them:
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration of that pass x clock of that pass)
  clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / duration of the pass that counted it
(the SQ pass carries no GRBM counter of its own: its busy fraction is quoted against SQ_BUSY_CYCLES — chip-wide cycles in which
any wave was resident, per SE — scaled by the duration ratio, and against the GRBM pass's clock as a second opinion)."""
import csv
import re
import sys
from collections import defaultdict

N_SIMD = 1024  # 256 CUs x 4
N_XCD = 8      # GRBM_GUI_ACTIVE is summed over the 8 XCDs


def short(name):
    m = re.match(r"void (\w+)<", name)
    base = m.group(1) if m else name.split("(")[0]
    if base == "tail_kernel":  # its template arguments select different kernels (4- / 8-wave, LDS-resident FFN operands, ...): keep them
        base = re.match(r"void (tail_kernel<[^>]*>)", name).group(1).replace(" ", "")
    for tag in ("EpiEmbed", "EpiOut", "EpiQK") + (("EpiTiled", "EpiResLN") if "layer_tail" not in name else ()):  # (layer_tail*: one tag per kernel)
        if tag in name:
            base += ":" + tag
    return base


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> durations (us) of the dispatches that counted it
    for p in paths:
        for row in csv.DictReader(open(p)):
            k, c = short(row["Kernel_Name"]), row["Counter_Name"]
            acc[k][c].append(float(row["Counter_Value"]))
            if row.get("End_Timestamp") and row.get("Start_Timestamp"):
                dur[k][c].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    counters = sorted({c for k in acc.values() for c in k})
    print("kernel," + ",".join(counters) + ",launches,us_grbm_pass,us_sq_pass,clock_ghz_grbm_pass,mfma_busy_frac_at_that_clock")
    for k, d in sorted(acc.items()):
        if k.startswith("void at::") or k.startswith("__amd"):  # torch / runtime helpers of bench.py's setup
            continue
        n = max(len(v) for v in d.values())
        mean = {c: sum(v) / len(v) for c, v in d.items()}
        us_g = sum(dur[k]["GRBM_GUI_ACTIVE"]) / len(dur[k]["GRBM_GUI_ACTIVE"]) if dur[k].get("GRBM_GUI_ACTIVE") else None
        us_s = sum(dur[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(dur[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) if dur[k].get("SQ_VALU_MFMA_BUSY_CYCLES") else None
        clock = mean["GRBM_GUI_ACTIVE"] / N_XCD / us_g / 1e3 if us_g else None
        busy = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD / (us_s * 1e3 * clock) if (us_s and clock and "SQ_VALU_MFMA_BUSY_CYCLES" in mean) else None
        print(('"' + k + '"' if "," in k else k) + "," + ",".join(f"{mean[c]:.4g}" if c in mean else "" for c in counters) + f",{n}," +
              ",".join("" if v is None else f"{v:.4g}" for v in (us_g, us_s, clock, busy)))


if __name__ == "__main__":
    main(sys.argv[1:])
