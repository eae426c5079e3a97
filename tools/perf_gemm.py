#!/usr/bin/env python3
"""Microbenchmark of ibl_linear_f16 on the four ViT-B/14 layer shapes (224 crops x 257 tokens)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMPS = "--stamps" in sys.argv
if STAMPS:                                   # `make -C instance-based-loc_amd/csrc lab` first
    sys.argv.remove("--stamps")
    os.environ.setdefault("IBLOC_LIB", os.path.join(ROOT, "tools", "_lab", "libibloc_lab.so"))
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibloc_amd import vit as V

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 224 * 257
shapes = [("qkv", 2304, 768, 0), ("proj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)]
tot_ms, tot_fl = 0.0, 0.0
for name, n_out, n_in, epi in shapes:
    x = torch.randn(rows, n_in, device="cuda").to(torch.float16)
    W = (torch.randn(n_out, n_in, device="cuda") / n_in ** 0.5).to(torch.float16)
    b = torch.randn(n_out, device="cuda")
    out = torch.zeros(rows, n_out, device="cuda", dtype=torch.float32 if epi == 2 else torch.float16)
    for _ in range(3):
        V.linear_f16(x, W, b, epi, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        V.linear_f16(x, W, b, epi, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * rows * n_out * n_in
    tot_ms += ms
    tot_fl += fl
    print(f"{name:5s} N={n_out:5d} K={n_in:5d} epi={epi}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    if STAMPS:
        from ibloc_amd import _lib
        nb = min(8192, ((rows + 255) // 256) * (n_out // 256))
        buf = np.zeros(5 * nb, dtype=np.int64)
        _lib.lib.ibl_gemm_stamps_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_int(5 * nb))
        st = buf.reshape(nb, 5)
        t0 = st[:, 0].min()
        d = np.diff(st[:, :4], axis=1).astype(np.float64)            # s_memtime ticks
        print(f"      blocks {nb}: prologue {d[:, 0].mean():.0f}, k-loop {d[:, 1].mean():.0f}, epilogue+drain "
              f"{d[:, 2].mean():.0f} clocks per block; "
              f"k-loop min/max {d[:, 1].min():.0f}/{d[:, 1].max():.0f}")
print(f"layer total {tot_ms * 1e3:.1f} us  {tot_fl / tot_ms / 1e9:.1f} TFLOP/s  (x12 = {tot_ms * 12:.2f} ms)")
