#!/usr/bin/env python3
"""The large GEMM shapes of the path with the split-bf16 products on and off: duration of each and the error of both
against an fp64 product of the same fp32 operands (max |error| and rms error, in units of 2^-24 * sqrt(K) * rms(a) *
rms(b): one fp32 rounding per accumulated product)."""
import math
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ick_amd  # noqa: E402,F401
from ick_amd import ops  # noqa: E402


def run(name, M, N, K, akm, bkm, split_k=1, reps=30, a_grp=0):
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + 7 * K)
    A = (torch.randn(K, M, device="cuda", generator=g) if akm else torch.randn(M, K, device="cuda", generator=g))
    B = (torch.randn(K, N, device="cuda", generator=g) if bkm else torch.randn(N, K, device="cuda", generator=g)) * 0.1
    ref = (A.double().t() if akm else A.double()) @ (B.double() if bkm else B.double().t())
    unit = 2.0 ** -24 * math.sqrt(K) * 0.1
    row = [name]
    for on in MODES:
        ops.set_gemm_split(2 * on)
        out = torch.zeros(M, N, device="cuda")

        def call():
            ops.gemm_raw(A, B, out, M, N, K, 1 if akm else K, M if akm else 1, 1 if bkm else K, N if bkm else 1, N,
                         atomic=split_k > 1, split_k=split_k)
        call()
        torch.cuda.synchronize()
        err = (out.double() - ref)
        emax, erms = err.abs().max().item() / unit, err.pow(2).mean().sqrt().item() / unit
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        row.append("%s %7.1f us %6.1f TF err max %5.2f rms %5.3f" % ("split" if on else "exact", us,
                                                                     2.0 * M * N * K / us / 1e6, emax, erms))
    print(" | ".join(row), flush=True)


import os  # noqa: E402
MODES = (1,) if os.environ.get("SPLIT_ONLY") else ((0,) if os.environ.get("EXACT_ONLY") else (0, 1))
shapes = [("conv1 12544x300x2048 A k-major   ", 12544, 300, 2048, True, False, 1),
          ("cross K/V 13824x1800x300         ", 13824, 1800, 300, False, False, 1),
          ("vocab fwd 1280x10000x300         ", 1280, 10000, 300, False, False, 1),
          ("vocab dgrad 1280x300x10000 split9", 1280, 300, 10000, False, True, 9),
          ("vocab wgrad 10000x300x1280 split2", 10000, 300, 1280, True, True, 2),
          ("K/V wgrad 600x300x13824 split16  ", 600, 300, 13824, True, True, 16),
          ("K/V cfg4 25000x1800x300          ", 25088, 1800, 300, False, False, 1),
          ("square 4096x4096x2048            ", 4096, 4096, 2048, False, False, 1)]
if os.environ.get("WGRAD_ONLY"):
    sk = int(os.environ.get("WGRAD_SPLIT", "0"))
    shapes = [("vocab wgrad 10000x300x1280 ", 10000, 300, 1280, True, True, sk or 2),
              ("K/V wgrad 600x300x13824    ", 600, 300, 13824, True, True, sk or 16),
              ("linear1 wgrad 1024x300x1280", 1024, 300, 1280, True, True, sk or 2),
              ("vocab dgrad 1280x300x10000 ", 1280, 300, 10000, False, True, sk or 9)]
for s in shapes:
    run(*s)
ops.set_gemm_split(0)
