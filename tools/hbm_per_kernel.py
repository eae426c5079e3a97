#!/usr/bin/env python3
"""Achieved HBM bandwidth per kernel family of one bench step run alone: bytes from the FETCH_SIZE / WRITE_SIZE PMC passes
(profiles/rNN/pmc/*_counter_collection_ibl_kernels.csv, corrected as in tools/pmc_summary.py), time from the un-profiled kernel trace
(profiles/rNN/bench_default_last_step_breakdown.txt).

    python tools/hbm_per_kernel.py profiles/r01 > profiles/r01/hbm_per_kernel.txt"""
import csv
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def last_step_bytes(path, counter, scale):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", "").strip(), float(r["Counter_Value"])))
    rows.sort()
    start = max(i for i, k, _ in rows if "ibl_resample_h" in k)
    out = defaultdict(float)
    for i, k, v in rows:
        if i >= start:
            out[k] += v * scale
    return out


fetch = last_step_bytes(f"{root}/pmc/fetch_counter_collection_ibl_kernels.csv", "FETCH_SIZE", 1024.0 * 2.0)
write = last_step_bytes(f"{root}/pmc/write_counter_collection_ibl_kernels.csv", "WRITE_SIZE", 1024.0)
times = {}
for line in open(f"{root}/bench_default_last_step_breakdown.txt"):
    m = re.match(r"(?:void )?(\S.*?)\s+(\d+)\s+([\d.]+) ms$", line.strip())
    if m:
        times[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
print("# one bench step run alone (C2 workload): HBM bytes from the PMC passes, time from the un-profiled kernel trace; peak 8000 GB/s")
print(f"{'kernel':56s} {'calls':>5s} {'ms':>7s} {'read MB':>9s} {'write MB':>9s} {'GB/s':>7s} {'of peak':>8s}")
for k, (n, ms) in sorted(times.items(), key=lambda kv: -kv[1][1]):
    key = next((q for q in fetch if q.startswith(k) or k.startswith(q)), None)
    if key is None:
        continue
    rb, wb = fetch.get(key, 0.0), write.get(key, 0.0)
    gbs = (rb + wb) / (ms * 1e-3) / 1e9
    print(f"{k[:56]:56s} {n:5d} {ms:7.2f} {rb / 1e6:9.1f} {wb / 1e6:9.1f} {gbs:7.0f} {gbs / 8000:8.3f}")
