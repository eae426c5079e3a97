"""In-process kernel timing for bench.py's roofline line.

The library brackets selected kernel families with HIP events recorded on the stream the kernels are launched
on (csrc/common.cpp, ibl_prof_*); this module reads the accumulated device time and algorithmic work."""
import ctypes as C

from . import _lib

GEMM, ATTN, SPFH = 1, 2, 3
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
