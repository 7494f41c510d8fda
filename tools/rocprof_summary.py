#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (`--kernel-trace --stats`) as a small CSV:
kernel, calls, total_us, avg_us, pct.   usage: rocprof_summary.py results.db out.csv"""
import csv
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"smpc\d+(\w+?_body)", name)
    if m:
        return m.group(1)
    return name[:80]


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([short(name), calls, "%.1f" % (tot / 1e0 if tot < 1e12 else tot), "%.2f" % avg, "%.3f" % pct])
    print("wrote", out)


if __name__ == "__main__":
    main()
