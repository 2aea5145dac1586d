"""Where do the workgroups of a large GEMM spend their time?  Builds a copy of the library with -DICK_GEMM_STAMPS, runs
one problem and prints, for the workgroups that ran on one CU, the timeline (shader-clock ticks -> us): start, first LDS
stage ready, K loop done, end.  usage: gemm_stamps.py [M N K [split_mode [kmajor [split_k]]]]"""
import collections
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ick_amd.build as b  # noqa: E402
import ick_amd.lib as L  # noqa: E402

dbg = os.path.join(ROOT, "gpurun_out", "libick_amd_gstamps.so")
subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DICK_GEMM_STAMPS", "-shared", "-o", dbg] + b.sources())
L.LIB_PATH = dbg
import torch  # noqa: E402
import ick_amd  # noqa: E402,F401
from ick_amd import ops  # noqa: E402

TICKS_PER_US = 2400.0     # s_memtime runs at the shader clock on this part (measured: a workgroup of 22 us = 54 k ticks)
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (13824, 1800, 300)
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
km = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # 1: both operands k-major (weight-gradient form)
split_k = int(sys.argv[6]) if len(sys.argv) > 6 else 1
ops.set_gemm_split(mode)
out = torch.zeros(M, N, device="cuda")
if km:
    A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
    for _ in range(3):
        ops.gemm_raw(A, B, out, M, N, K, 1, M, 1, N, N, atomic=True, split_k=split_k)
else:
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    for _ in range(3):
        ops.gemm_raw(A, B, out, M, N, K, K, 1, K, 1, N)
torch.cuda.synchronize()
n = 16384
buf = (ctypes.c_ulonglong * (8 * n))()
lib = ctypes.CDLL(dbg)
assert lib.ick_debug_read_gemm_stamps(buf, n) == 0
rows = [[buf[8 * i + j] for j in range(8)] for i in range(n)]
rows = [r for r in rows if r[0] and r[3]]
t0 = min(r[0] for r in rows)
tend = max(r[3] for r in rows)
print("%d workgroups recorded, span %.1f us" % (len(rows), (tend - t0) / TICKS_PER_US))
by_cu = collections.defaultdict(list)
for r in rows:
    hw, xcc = r[7] & 0xffffffff, r[7] >> 32
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    by_cu[(xcc & 0xf, se, sh, cu)].append(r)
print("%d distinct CUs; workgroups per CU min/max %d/%d" % (len(by_cu), min(map(len, by_cu.values())),
                                                           max(map(len, by_cu.values()))))
key = sorted(by_cu)[len(by_cu) // 2]
print("CU", key, ": start, +prologue, +loop, +epilogue (us)")
for r in sorted(by_cu[key]):
    print("  %8.2f  %6.2f %6.2f %6.2f   end %8.2f" % ((r[0] - t0) / TICKS_PER_US, (r[1] - r[0]) / TICKS_PER_US, (r[2] - r[1]) / TICKS_PER_US,
                                                     (r[3] - r[2]) / TICKS_PER_US, (r[3] - t0) / TICKS_PER_US))
pro = sum(r[1] - r[0] for r in rows) / len(rows) / TICKS_PER_US
loop = sum(r[2] - r[1] for r in rows) / len(rows) / TICKS_PER_US
epi = sum(r[3] - r[2] for r in rows) / len(rows) / TICKS_PER_US
addr = sum(r[4] - r[2] for r in rows) / len(rows) / TICKS_PER_US
print("mean over all workgroups: prologue %.2f us, K loop %.2f us, epilogue %.2f us (of which %.2f us before the first "
      "store: bias / offsets)" % (pro, loop, epi, addr))
