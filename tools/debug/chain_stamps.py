"""Phase timing of the row-chain kernel (diagnostic): builds a copy of the library with -DICK_CHAIN_STAMPS, runs one
chain shape and prints, for thread 0 of workgroup 0 of the last launch, the shader-clock deltas between the phase
stamps (s_memtime: shader-clock cycles)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ick_amd.build as b
import ick_amd.lib as L
dbg = os.path.join(ROOT, "gpurun_out", "libick_amd_dbg.so")
subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DICK_CHAIN_STAMPS", "-shared", "-o", dbg] + b.sources())
L.LIB_PATH = dbg
import torch
from ick_amd import ops
lib = ctypes.CDLL(dbg)
names = ["prefetch issue", "A rows -> LDS issue", "barrier", "GEMM 1", "partials + barrier", "LayerNorm", "barrier",
         "GEMM 2", "partials + barrier", "epilogue"]
for M, K1, N2, heads in [(1280, 300, 300, True), (1280, 300, 512, False), (1280, 512, 900, True), (1280, 300, 0, False)]:
    d, H, T = 300, 10, 20
    a = torch.randn(M, K1, device="cuda"); res = torch.randn(M, d, device="cuda")
    w1 = ops.pack_weight(torch.randn(d, K1, device="cuda")); b1 = torch.randn(d, device="cuda")
    g = torch.ones(d, device="cuda"); be = torch.zeros(d, device="cuda")
    x = torch.empty(M, d, device="cuda"); o = torch.empty(M, d, device="cuda")
    w2 = ops.pack_weight(torch.randn(N2, d, device="cuda")) if N2 else None
    b2 = torch.randn(N2, device="cuda") if N2 else None
    y2 = None
    hd = None
    if N2:
        if heads:
            y2 = torch.empty(M // T, N2 // d, H, T, ops.DHP, device="cuda"); hd = (N2 // d, H, T, 0, T)
        else:
            y2 = torch.empty(M, N2, device="cuda")
    for _ in range(5):
        ops.rowchain_fwd(a, w1, b1, res, g, be, 1e-5, x, o_out=o, save_stats=True, w2p=w2, b2=b2, y2=y2, heads=hd)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert lib.ick_debug_read_chain_stamps(buf) == 0
    n = 10 if N2 else 6
    t = [buf[i] for i in range(n + 1)]
    # s_memtime counts shader-clock cycles on gfx950 (~2.1 GHz while these kernels run; csrc/gemm_ps.hip's stamps measure the
    # clock against s_memrealtime), not the 100 MHz of the reference clock: print cycles and shares
    tot = t[-1] - t[0]
    print("K1=%d N2=%d: %d cycles inside workgroup 0 (~%.1f us at 2.1 GHz): " % (K1, N2, tot, tot / 2100.0) +
          ", ".join("%s %d (%.0f %%)" % (names[i], t[i + 1] - t[i], 100.0 * (t[i + 1] - t[i]) / tot) for i in range(n)))
