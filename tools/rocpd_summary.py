#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
    python tools/rocpd_summary.py gpurun_out/prof/xxx_results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                       "group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["%-90s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "share")]
    for n, c, t, a, mn, mx in rows:
        short = n if len(n) <= 90 else n[:87] + "..."
        lines.append("%-90s %8d %14.1f %12.2f %12.2f %12.2f %6.2f%%" % (short, c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
