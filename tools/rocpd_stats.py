#!/usr/bin/env python3
"""Per-kernel time summary (the columns of rocprofv3 --stats) from a rocprofv3 rocpd database.

    python tools/rocpd_stats.py gpurun_out/prof_t196/t196_results.db > profiles/r02_step_b256_t196_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), round(avg(d.end-d.start),1), min(d.end-d.start), "
                      f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], r[3], round(100 * r[2] / tot, 3), r[4], r[5]])


if __name__ == "__main__":
    main(sys.argv[1])
