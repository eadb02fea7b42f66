"""Turns a rocprofv3 result database (rocpd .db, the default output of ROCm 7.2's rocprofv3
--kernel-trace --stats) into the per-kernel summary CSV that is committed under profiles/.

    python profiles/summarize.py gpurun_out/prof1/r1_results.db profiles/r01_bench_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "average_us", "percent"])
        for name, calls, total, avg, pct in rows:
            w.writerow([name.split("(")[0], calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.3f}"])
    for r in rows[:8]:
        print(f"{r[0].split('(')[0]:40s} calls={r[1]:5d} avg={r[3]:10.1f} us  {r[4]:5.1f} %")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
