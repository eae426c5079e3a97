#!/usr/bin/env python3
"""LDS counters of the GEMM kernels from a rocprofv3 --pmc pass of tools/perf_gemm.py (SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES): measured 0 bank-conflict cycles for every
epilogue's swizzle and 3 % of the wave cycles in LDS issue stalls.

    python tools/pmc_lds_gemm.py <counter_collection.csv>"""
import csv, sys
from collections import defaultdict
rows=defaultdict(lambda: defaultdict(float)); 
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if "gemm" not in k: continue
    rows[k][r["Counter_Name"]]+=float(r["Counter_Value"]); rows[k]["n_"+r["Counter_Name"]]+=1
for k,c in rows.items():
    print(k, {n:round(v) for n,v in c.items() if not n.startswith("n_")})
    if c.get("SQ_LDS_IDX_ACTIVE"): print("   bank conflict / idx active =", c["SQ_LDS_BANK_CONFLICT"]/c["SQ_LDS_IDX_ACTIVE"])
