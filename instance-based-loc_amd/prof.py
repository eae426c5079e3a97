"""In-process kernel timing for bench.py's roofline line.

The library brackets selected kernel families with HIP events recorded on the stream the kernels are launched
on (csrc/common.cpp, ibl_prof_*); this module reads the accumulated device time and algorithmic work."""
import ctypes as C

from . import _lib

GEMM, ATTN, SPFH = 1, 2, 3
ST_FEATURES, ST_FEATMATCH, ST_RANSAC, ST_ICP, ST_EVAL, ST_OUTLIER = 10, 11, 12, 13, 14, 15
PEAK_F32_VALU_TFLOPS = 157.3  # fp32 vector peak, the bound SURVEY names for RANSAC
PEAK_F16_TFLOPS = 2500.0      # dense fp16 / bf16 MFMA (same rate), /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def reset(enable=True):
    _lib.check(_lib.lib.ibl_prof_enable(1 if enable else 0), "ibl_prof_enable")


def read(kid):
    ms, units, n = C.c_double(), C.c_double(), C.c_int64()
    _lib.check(_lib.lib.ibl_prof_read(kid, C.byref(ms), C.byref(units), C.byref(n)), "ibl_prof_read")
    return ms.value, units.value, n.value


def roofline(traffic=None):
    """Roofline object of the dominant kernel family (the fp16 GEMM of the ViT encoder).  `traffic` = HBM bytes per
    launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command (tools/pmc_summary.py)."""
    ms, flops, n = read(GEMM)
    if n == 0 or ms <= 0:
        return None
    achieved = flops / (ms * 1e-3) / 1e12
    return {"kernel": "ibl_gemm_f16_tn", "bound": "mfma", "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_F16_TFLOPS, "traffic": traffic, "launches": n, "avg_launch_us": ms * 1e3 / n,
            "flops_per_launch": flops / n}


# SURVEY 8d stage table: (family id, stage, bound, unit of `achieved`, peak, divisor of units / s -> unit, what one unit is)
_STAGES = [
    (ST_OUTLIER, "outlier (a9)", "hbm", "GB/s", PEAK_HBM_GBS, 1e9, "13 B per detected point (12 in + 1 mask byte out)"),
    (ST_FEATURES, "normals+FPFH (a11)", "hbm", "GB/s", PEAK_HBM_GBS, 1e9, "444 B per point (normals 24 + SPFH 156 + FPFH 264), compulsory"),
    (ST_FEATMATCH, "feature match (a12)", "mfma", "TFLOP/s", PEAK_F16_TFLOPS, 1e12, "2*2*33*Ns*Nt FLOP per job (both directions, every source x target pair)"),
    (ST_RANSAC, "RANSAC (a12)", "valu", "Mhyp/s", None, 1e6, "hypotheses walked (reference criteria 4e6 / 0.99, confidence exit on)"),
    (ST_ICP, "coloured ICP (a12)", "hbm", "GB/s", PEAK_HBM_GBS, 1e9, "56 B per source point and iteration run"),
    (ST_EVAL, "evaluate (a13)", "hbm", "GB/s", PEAK_HBM_GBS, 1e9, "24 B per detected point and candidate"),
]


def stage_rooflines():
    """One entry per registration stage of SURVEY 8d: device time of the stage's launches (HIP events on their stream) and the
    algorithmic work they did, since the last reset()."""
    out = []
    for kid, name, bound, unit, peak, div, what in _STAGES:
        ms, units, n = read(kid)
        if n == 0 or ms <= 0:
            continue
        ach = units / (ms * 1e-3) / div
        out.append({"stage": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": (ach / peak) if peak else None,
                    "ms": ms, "calls": n, "work": units, "work_unit": what})
    return out
