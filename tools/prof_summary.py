"""Summarise a rocprofv3 results DB (rocpd sqlite, the ROCm 7.2 default output) into the
per-kernel table that `--stats` prints: calls, total/avg/min/max duration, share.

    python tools/prof_summary.py gpurun_out/prof/r01_results.db profiles/r01_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, n, tot, avg, mn, mx in rows:
            w.writerow([name, n, tot, round(avg, 1), round(100.0 * tot / total, 3), mn, mx])
    print("%d kernels, %d dispatches, %.3f ms total -> %s" % (len(rows), sum(r[1] for r in rows), total / 1e6, out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
