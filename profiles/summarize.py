"""Turns the kernel statistics of `rocprofv3 --kernel-trace --stats --output-format csv` (<name>_kernel_stats.csv)
into the per-kernel summary CSV that is committed under profiles/ (kernel names without their argument lists).

    python profiles/summarize.py gpurun_out/prof/b_kernel_stats.csv profiles/r02_a_bench_kernel_stats.csv
"""
import csv
import sys


def main(src, out):
    rows = list(csv.DictReader(open(src)))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "average_us", "percent"])
        for r in rows:
            name = r["Name"].split("(")[0].replace("void ", "")
            w.writerow([name, r["Calls"], f"{float(r['TotalDurationNs']) / 1e3:.3f}", f"{float(r['AverageNs']) / 1e3:.3f}", f"{float(r['Percentage']):.3f}"])
    for r in rows[:8]:
        print(f"{r['Name'].split('(')[0]:40s} calls={int(r['Calls']):5d} avg={float(r['AverageNs']) / 1e3:10.1f} us  {float(r['Percentage']):5.1f} %")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
