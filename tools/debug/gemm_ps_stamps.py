"""Where does a wave of the pre-split GEMM (csrc/gemm_ps.hip) spend a slice?  Builds a copy of the library with
-DICK_PS_STAMPS, runs one shape per tile and prints wave 0's mean shader-clock ticks per slice in each phase:
DMA issue | B fragment requests + A fragment read + split | MFMA issue | wait for the next slice's DMA | barrier.
usage: gemm_ps_stamps.py [shape [tile ...]]   (shapes of tools/gemm_ps_bench.py)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import ick_amd.build as b
    import ick_amd.lib as L
    dbg = os.path.join(ROOT, "gpurun_out", "libick_amd_psstamps.so")
    L.LIB_PATH = dbg
    import torch
    import ick_amd  # noqa: F401
    from ick_amd import ops
    from gemm_ps_bench import SHAPES
    name = sys.argv[2]
    M, N, K, lay, split_k = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(1)
    if lay == "nchw":
        A = torch.randn(M // 196, K, 196, device="cuda", generator=g); aargs = (1, 196); akw = dict(a_grp=196, a_gs=K * 196)
    elif lay.startswith("kmaj"):
        ld = int(lay[4:])
        A = torch.randn(K, ld, device="cuda", generator=g)[:, :M]; aargs = (1, ld); akw = {}    # element (m, k) at A[k, m]
    else:
        A = torch.randn(M, K, device="cuda", generator=g); aargs = (K, 1); akw = {}
    if split_k > 1:
        akw.update(atomic=True, split_k=split_k)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.1
    ps = ops.presplit_buffer(N, K, "cuda")
    ops.presplit_weights([(W, ps)])
    out = torch.zeros(M, N, device="cuda")
    ops.set_gemm_split(1)
    for _ in range(5):
        ops.gemm_raw(A, W, out, M, N, K, *aargs, K, 1, N, b_ps=ps, **akw)
    torch.cuda.synchronize()
    info = L.GemmPlanInfo()
    L.check(L.load().ick_gemm_plan(ops.gemm_args(A, W, out, M, N, K, *aargs, K, 1, N, b_ps=ps, **akw), info), "plan")
    buf = (ctypes.c_ulonglong * 10)()
    assert ctypes.CDLL(dbg).ick_debug_read_ps_stamps(buf) == 0
    n = max(1, buf[5])
    names = ["DMA issue", "B requests + A read + split", "MFMA issue", "wait next slice's DMA", "barrier"]
    tot = sum(buf[i] for i in range(5)) / n
    print("%-10s tile %3dx%-3d: %5.0f ticks per slice = " % (name, info.tile_m, info.tile_n, tot) +
          " | ".join("%s %4.0f" % (nm, buf[i] / n) for i, nm in enumerate(names)) +
          "  ||  per tile: prologue %d, %d slices = %d, epilogue %d ticks; in-kernel clock %.2f GHz (%d ticks in %.2f us)"
          % (buf[6], n, tot * n, buf[7], buf[8] / max(1, buf[9]) * 0.1, buf[8], buf[9] / 100.0), flush=True)
    sys.exit(0)

import ick_amd.build as b  # noqa: E402
dbg = os.path.join(ROOT, "gpurun_out", "libick_amd_psstamps.so")
os.makedirs(os.path.dirname(dbg), exist_ok=True)
subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DICK_PS_STAMPS", "-w", "-shared", "-o", dbg] + b.sources())
shape = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--shape" else (sys.argv[1] if len(sys.argv) > 1 else "conv1")
tiles = sys.argv[2:] if len(sys.argv) > 2 else ["2", "0", "6"]
for t in tiles:
    env = dict(os.environ, ICK_PS_TILE=t)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", shape], env=env, capture_output=True, text=True)
    sys.stdout.write(r.stdout if r.returncode == 0 else "FAILED tile %s: %s\n" % (t, r.stderr[-400:]))
