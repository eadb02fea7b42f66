"""Per-kernel sums of the counters collected by profiles/pmc_passes.sh.

    python profiles/pmc_summary.py gpurun_out/pmc1 [out.csv]
"""
import collections
import csv
import glob
import os
import sys


def main(d, out=None):
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for path in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k][row["Counter_Name"]] += 1
    dur = collections.defaultdict(float)
    ncall = collections.defaultdict(int)
    kt = sorted(glob.glob(os.path.join(d, "sq1_kernel_trace.csv")))
    if kt:
        with open(kt[0]) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                ncall[k] += 1
    rows = []
    for k in sorted(sums, key=lambda k: -dur.get(k, 0)):
        r = {"kernel": k, "calls": ncall.get(k, 0), "total_us": round(dur.get(k, 0), 1)}
        r.update({c: v for c, v in sorted(sums[k].items())})
        rows.append(r)
    cols = ["kernel", "calls", "total_us"] + sorted({c for r in rows for c in r} - {"kernel", "calls", "total_us"})
    w = csv.DictWriter(open(out, "w", newline="") if out else sys.stdout, fieldnames=cols)
    w.writeheader()
    for r in rows:
        w.writerow(r)
    return rows


if __name__ == "__main__":
    main(*sys.argv[1:])
