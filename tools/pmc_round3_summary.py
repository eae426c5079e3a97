#!/usr/bin/env python3
"""Summaries of the round-3 PMC passes (tools/profile_round3_c.sh):  python tools/pmc_round3_summary.py <pmc dir> <out dir>
  hbm_per_kernel.txt   per kernel: HBM bytes per launch (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024: rocprofv3 reports KB, and on gfx950 a
                       wide streaming read is tallied at half its bytes -- MI355X_MICROARCH.md, HBM section), mean duration from the
                       UN-profiled kernel trace of the same command, achieved GB/s, fraction of the 8 TB/s peak
  gemm_pmc.json        the GEMM's traffic per launch next to its algorithmic bytes (bench.py's roofline.traffic_from_profile)
  wave_split.json      per kernel: MFMA / VALU utilisation and the split of wave time into waiting / issuing (SQ_WAIT_ANY etc.)"""
import csv
import json
import sys
from collections import defaultdict

pmc, out = sys.argv[1], sys.argv[2]


def short(n):
    return n.split("(")[0].replace("void ", "").strip()


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path, newline="")):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations(path):
    d = {}
    for r in csv.DictReader(open(path, newline="")):
        d[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    return d


lines = []
gemm = {}
for tag, title in (("vit", "encoder forward of one 224-crop batch (tools/perf_vit.py dinov2_vitb14 224)"),
                   ("reg", "stage B of bench steps on a 300-instance memory (tools/perf_register.py: 224 detections of ~5 000 points per step)")):
    f = counters(f"{pmc}/{tag}_fetch_counter_collection_ibl_kernels.csv")
    w = counters(f"{pmc}/{tag}_write_counter_collection_ibl_kernels.csv")
    dur = durations(f"{pmc}/{tag}_kernel_stats.csv")
    lines.append(f"# {title}")
    lines.append(f"{'kernel':72s} {'launches':>8s} {'MB/launch':>10s} {'us/launch':>10s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
    rows = []
    for k in f:
        fb = sum(f[k]["FETCH_SIZE"]) * 1024.0 * 2.0 / max(1, len(f[k]["FETCH_SIZE"]))
        wb = sum(w[k]["WRITE_SIZE"]) * 1024.0 / max(1, len(w[k]["WRITE_SIZE"])) if k in w else 0.0
        if k not in dur:
            continue
        us, calls = dur[k]
        gbs = (fb + wb) / (us * 1e-6) / 1e9
        rows.append((us * calls, k, calls, (fb + wb) / 1e6, us, gbs))
        if "ibl_gemm_f16_tn" in k:
            g = gemm.setdefault("all", [0.0, 0.0, 0])
            g[0] += (fb + wb) * calls
            g[1] += us * calls
            g[2] += calls
    for _, k, calls, mb, us, gbs in sorted(rows, reverse=True):
        lines.append(f"{k[:72]:72s} {calls:8d} {mb:10.2f} {us:10.1f} {gbs:8.0f} {gbs / 8000:9.3f}")
    lines.append("")
open(f"{out}/hbm_per_kernel.txt", "w").write("\n".join(lines) + "\n")
if gemm:
    tot_b, tot_us, n = gemm["all"]
    # algorithmic bytes of the four layer GEMMs of ViT-B/14 at 57 568 rows (DESIGN (d)): qkv 332 MB, proj 415, fc1 415, fc2 664, x 11 full
    # blocks; the two-term blocks read their extra operand terms on top (counted as traffic, not as algorithmic bytes)
    algo = (332 + 415 + 415 + 664) * 1e6 * 11 / (4 * 11)
    json.dump({"kernel": "ibl_gemm_f16_tn (all epilogues, mean over the launches of one encoder forward)", "launches": n,
               "traffic_bytes_per_launch": tot_b / n, "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": tot_b / n / algo,
               "mean_launch_us": tot_us / n, "achieved_hbm_GBps": tot_b / (tot_us * 1e-6) / 1e9,
               "how": "FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 from separate rocprofv3 --pmc passes; durations from the un-profiled kernel trace",
               # the kernel source these counters were collected on: bench.py reports the figure as roofline.traffic only for this source
               "source_sha256_vit_hip": __import__("hashlib").sha256(open(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(
                   __import__("os").path.abspath(__file__))), "instance-based-loc_amd", "csrc", "vit.hip"), "rb").read()).hexdigest()},
              open(f"{out}/gemm_pmc.json", "w"), indent=1)
split = {}
for tag, path in (("vit", f"{pmc}/vit_mfma_counter_collection_ibl_kernels.csv"), ("reg", f"{pmc}/reg_wave_counter_collection_ibl_kernels.csv")):
    c = counters(path)
    for k, v in c.items():
        s = {n: sum(x) for n, x in v.items()}
        cyc = s.get("GRBM_GUI_ACTIVE", 0.0) / 8.0            # rocprofv3 sums the counter over the 8 XCDs
        wave = max(s.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        if cyc <= 0:
            continue
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md, PMC units)
        e = {"launches": len(v.get("GRBM_GUI_ACTIVE", [])), "wait_any_share": s.get("SQ_WAIT_ANY", 0.0) / wave,
             "wait_inst_share": s.get("SQ_WAIT_INST_ANY", 0.0) / wave, "active_inst_share": s.get("SQ_ACTIVE_INST_ANY", 0.0) / wave,
             "mean_waves_per_simd": 4.0 * wave / (cyc * 1024.0)}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in s:
            e["mfma_util"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
        if "SQ_ACTIVE_INST_VALU" in s:
            e["valu_util"] = 4.0 * s["SQ_ACTIVE_INST_VALU"] / (cyc * 1024.0)
        if "SQ_INSTS_VALU" in s:
            e["valu_insts_per_launch"] = s["SQ_INSTS_VALU"] / max(1, e["launches"])
        split[k] = e
json.dump(split, open(f"{out}/wave_split.json", "w"), indent=1, sort_keys=True)
print(open(f"{out}/hbm_per_kernel.txt").read())
