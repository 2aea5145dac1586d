#!/usr/bin/env python3
"""The four large GEMMs of the path on the pre-split kernel (csrc/gemm_ps.hip), per tile shape and A-load policy:
duration (events around `reps` back-to-back launches, operands as cold as a 256 MB Infinity Cache leaves them) and the
maximum difference from the exact fp32 MFMA path.  Every (tile, nt) setting runs in a child process: the plan reads
ICK_PS_TILE once.   python tools/gemm_ps_bench.py [shape-name-prefix]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {  # name: (M, N, K, a_layout, split_k)
    "conv1": (12544, 300, 2048, "nchw", 1),
    "conv1_b32": (6272, 300, 2048, "nchw", 1),            # cfg5's prefill
    "vocab_dgrad": (1280, 300, 10000, "rows", 0),
    "cross_kv": (12544, 1800, 300, "rows", 1),
    "vocab_fwd": (1280, 10000, 300, "rows", 1),
    "kv_wgrad": (600, 300, 13824, "kmaj1800", 24),      # dW_kv of one layer: A = a 600-column slice of the (13824, 1800) K/V gradient
    "kv_wgrad_s16": (600, 300, 13824, "kmaj1800", 16),
    "kv_wgrad_s8": (600, 300, 13824, "kmaj1800", 8),
    "cross_kv_cfg4": (12544, 1800, 300, "rows", 1),
    "vocab_fwd_cfg4": (1280, 50000, 300, "rows", 1),
}


def child(name):
    import torch
    import ick_amd  # noqa: F401
    from ick_amd import ops
    import ick_amd.lib as L
    M, N, K, lay, split_k = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(1)
    if lay == "nchw":
        A = torch.randn(M // 196, K, 196, device="cuda", generator=g)
        aargs = (1, 196)
        akw = dict(a_grp=196, a_gs=K * 196)
    elif lay.startswith("kmaj"):
        ld = int(lay[4:])
        A = torch.randn(K, ld, device="cuda", generator=g)[:, :M]     # element (m, k) at A[k, m]
        aargs = (1, ld)
        akw = {}
    else:
        A = torch.randn(M, K, device="cuda", generator=g)
        aargs = (K, 1)
        akw = {}
    W = torch.randn(N, K, device="cuda", generator=g) * 0.1
    ps = ops.presplit_buffer(N, K, "cuda")
    ops.presplit_weights([(W, ps)])
    if split_k == 0:      # what linear_bwd picks for the pre-split data gradient
        split_k = max(1, min(16, K // 512, 256 // ((M + 63) // 64)))
    out = torch.zeros(M, N, device="cuda")
    junk = torch.empty(96 * 1024 * 1024, device="cuda")      # 384 MB: flushes L2 and the Infinity Cache between launches

    def call(b_ps):
        ops.gemm_raw(A, W, out, M, N, K, *aargs, K, 1, N, atomic=split_k > 1, split_k=split_k, b_ps=b_ps, **akw)

    ops.set_gemm_split(0)
    out.zero_(); call(None); exact = out.clone()
    ops.set_gemm_split(1)
    info = L.GemmPlanInfo()
    a = ops.gemm_args(A, W, out, M, N, K, *aargs, K, 1, N, atomic=split_k > 1, split_k=split_k, b_ps=ps, **akw)
    L.check(L.load().ick_gemm_plan(a, info), "plan")
    out.zero_(); call(ps)
    diff = (out - exact).abs().max().item()
    times = {}
    for cold in (False, True):
        ts = []
        for _ in range(12):
            if cold:
                junk.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call(ps); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        times[cold] = sorted(ts)[len(ts) // 2]
    print("%-15s tile %3dx%-3d presplit %d split_k %2d: hot %7.1f us  cold %7.1f us  (%5.1f TF fp32-eq cold)  max|diff to exact| %.2e"
          % (name, info.tile_m, info.tile_n, info.presplit, info.split_k, times[False],
             times[True], 2.0 * M * N * K / times[True] / 1e6, diff), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    for name in SHAPES:
        if not name.startswith(want):
            continue
        for tile in (os.environ.get("TILES", ",0,1,2,3,4,5,6,7,8,9,10").split(",")):
            if True:
                env = dict(os.environ)
                if tile:
                    env["ICK_PS_TILE"] = tile
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], env=env, capture_output=True,
                                   text=True, timeout=300)
                sys.stdout.write(("auto: " if tile == "" else "      ") + (r.stdout if r.returncode == 0 else "FAILED %s %s: %s\n" % (name, tile, r.stderr[-300:])))
                sys.stdout.flush()
