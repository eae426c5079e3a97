#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_kernels.py <dir-or-csv> [name filter]
Prints, per kernel name, the number of dispatches and the mean of every counter (per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    p = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "ibl_"
    files = [p] if os.path.isfile(p) else glob.glob(p + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f, newline="")):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if flt in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        print(f"{k[:90]}  ({n} dispatches)")
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"    {c:32s} {sum(v) / len(v):18.1f}")


if __name__ == "__main__":
    main()
