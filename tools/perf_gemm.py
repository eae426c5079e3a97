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

PAD = int(os.environ.get("GEMM_PAD", "0"))          # elements added to the row stride of both operands (L2 channel spreading experiment)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 224 * 257
shapes = [("qkv", 2304, 768, 0), ("proj", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)]
tot_ms, tot_fl = 0.0, 0.0
for name, n_out, n_in, epi in shapes:
    x = torch.randn(rows, n_in + PAD, device="cuda").to(torch.float16)[:, :n_in]
    W = (torch.randn(n_out, n_in + PAD, device="cuda") / n_in ** 0.5).to(torch.float16)[:, :n_in]
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
        buf = np.zeros(512 * 16 * 4, dtype=np.int64)
        _lib.lib.ibl_gemm_stamps_clear()
        V.linear_f16(x, W, b, epi, out=out)
        torch.cuda.synchronize()
        _lib.lib.ibl_gemm_stamps_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_int(buf.size))
        st = buf.reshape(512, 16, 4).astype(np.float64)
        used = st[:, :, 3] > 0                                            # (block, tile iteration) pairs that ran
        started = st[:, 0, 0] > 0
        t0 = st[:, 0, 0][started].min()
        tiles_per_block = used.sum(1)
        blocks = int((tiles_per_block > 0).sum())
        top = (st[:, :, 1] - st[:, :, 0])[used]                           # tile top: wait for the first stage (+ previous stores), barrier
        kl = (st[:, :, 2] - st[:, :, 1])[used]
        ep = (st[:, :, 3] - st[:, :, 2])[used]
        end = st[:, :, 3][used].max() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.lib.ibl_gemm_stamps_clear()
        e0.record(); V.linear_f16(x, W, b, epi, out=out); e1.record(); torch.cuda.synchronize()
        one_us = e0.elapsed_time(e1) * 1e3
        first = used.copy(); first[:, 1:] = False
        later = used & ~first
        print(f"      {blocks} blocks x {tiles_per_block[tiles_per_block > 0].mean():.2f} tiles; s_memtime ticks: kernel span {end:.0f} ticks = {one_us:.0f} us alone ({end / one_us / 1e3:.2f} ticks / ns); per tile: "
              f"top wait {top.mean():.0f} (first tile {(st[:, :, 1] - st[:, :, 0])[first].mean():.0f}, later {(st[:, :, 1] - st[:, :, 0])[later].mean() if later.any() else 0:.0f}), "
              f"K loop {kl.mean():.0f} (min {kl.min():.0f} max {kl.max():.0f}), prologue(next) + epilogue issue {ep.mean():.0f}; "
              f"block busy sum {(top.sum() + kl.sum() + ep.sum()) / blocks:.0f}")
print(f"layer total {tot_ms * 1e3:.1f} us  {tot_fl / tot_ms / 1e9:.1f} TFLOP/s  (x12 = {tot_ms * 12:.2f} ms)")
