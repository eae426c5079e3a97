#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE / SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY /
SQ_ACTIVE_INST_ANY into per-kernel-family MFMA utilisation and wave-time split, over the dispatches of the LAST bench step (the step
bench.py runs alone after the timed region).

    python tools/pmc_mfma_summary.py <counter_collection.csv> profiles/r01/gemm_mfma_pmc.json

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 256 CUs x 4 SIMDs) with kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums
the counter over the 8 XCDs; MI355X_MICROARCH.md, DVFS section).  SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES."""
import csv
import json
import sys
from collections import defaultdict

rows = defaultdict(dict)
names = {}
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        d = int(r["Dispatch_Id"])
        rows[d][r["Counter_Name"]] = float(r["Counter_Value"])
        names[d] = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
ids = sorted(rows)
last = max(i for i in ids if "ibl_resample_h" in names[i])              # first kernel of the last step
fam = defaultdict(lambda: defaultdict(float))
for i in ids:
    if i < last:
        continue
    k = names[i]
    key = "ibl_gemm_f16_tn (all epilogues)" if "ibl_gemm_f16_tn" in k else k
    for c, v in rows[i].items():
        fam[key][c] += v
    fam[key]["launches"] += 1
out = {}
for k, c in fam.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0:
        continue
    wave = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    out[k] = {"launches": int(c["launches"]), "kernel_cycles": cyc,
              "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 256 * 4),
              "wave_time_parked": c.get("SQ_WAIT_ANY", 0.0) / wave, "wave_time_issue_stall": c.get("SQ_WAIT_INST_ANY", 0.0) / wave,
              "wave_time_issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave}
top = dict(sorted(out.items(), key=lambda kv: -kv[1]["kernel_cycles"])[:14])
json.dump({"per_kernel_family_last_step": top,
           "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs); wave_time_* = SQ_WAIT_ANY | SQ_WAIT_INST_ANY | SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES"},
          open(sys.argv[2], "w"), indent=1)
for k, v in top.items():
    print(f"{k[:52]:52s} n={v['launches']:4d} mfma_util {v['mfma_util']:.3f} parked {v['wave_time_parked']:.2f} issue-stall {v['wave_time_issue_stall']:.2f} issuing {v['wave_time_issuing']:.2f}")
