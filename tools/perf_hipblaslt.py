#!/usr/bin/env python3
"""Calibration only (not product code): what the vendor library reaches on the four ViT-B/14 GEMM shapes (plain bf16 GEMM,
no fused epilogue), to tell a practical ceiling of this board from headroom in ibl_gemm_f16_tn."""
import sys
import torch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 224 * 257
for name, n_out, n_in in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("big", 8192, 8192)]:
    m = rows if name != "big" else 8192
    x = torch.randn(m, n_in, device="cuda").to(torch.bfloat16)
    W = (torch.randn(n_out, n_in, device="cuda") / n_in ** 0.5).to(torch.bfloat16)
    out = torch.empty(m, n_out, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(x, W.t(), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.matmul(x, W.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:5s} M={m} N={n_out} K={n_in}: {ms * 1e3:8.1f} us  {2.0 * m * n_out * n_in / ms / 1e9:7.1f} TFLOP/s")
