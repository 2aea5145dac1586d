#!/usr/bin/env python3
"""Yardstick: what the vendor library (rocBLAS / hipBLASLt through torch.matmul, exact fp32) reaches on the large GEMM
shapes of the path.  Not part of the product; printed next to tools/probes/probe_ops for DESIGN.md."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
shapes = [("conv1 [12544x2048]x[2048x300], A k-major", 12544, 300, 2048, True),
          ("cross K/V [12544x300]x[300x1800]", 12544, 1800, 300, False),
          ("vocab [1280x300]x[300x10000]", 1280, 10000, 300, False),
          ("4096x4096x2048", 4096, 4096, 2048, False),
          ("chain 1280x300x300", 1280, 300, 300, False)]
for name, M, N, K, akm in shapes:
    a = torch.randn(K, M, device="cuda").t() if akm else torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(5):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        torch.matmul(a, w.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print("%-45s %8.1f us  %6.1f TFLOP/s  (%.2f of 157.3)" % (name, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 157.3))
