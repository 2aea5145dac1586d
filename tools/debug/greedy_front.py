"""Kernel timeline of the prefill part of one greedy bench step (Encoder.conv1 ... first decode kernel) from a
rocprofv3 --kernel-trace CSV directory.  usage: greedy_front.py <dir>"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
    return m.group(1) if m else r["Kernel_Name"][:40]
# last step: from the last-but-one conv1 launch (k-major A: gemm_kernel<2, 2, 1, 1, true, false) to the next dec_self
starts = [i for i, r in enumerate(rows) if "gemm_kernel<2, 2, 1, 1, true, false" in r["Kernel_Name"]]
a = starts[-2]
b = next(i for i in range(a, len(rows)) if "dec_self_kernel" in rows[i]["Kernel_Name"])
end_step = next(i for i in range(b, len(rows)) if i + 1 == len(rows) or "gemm_kernel<2, 2, 1, 1, true, false" in rows[i + 1]["Kernel_Name"])
t0 = int(rows[a]["Start_Timestamp"])
busy = 0.0
for r in rows[a:b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    gap = s - busy if s > busy else 0.0
    busy = max(busy, e)
    print("%8.1f %8.1f %6.1f q%s %-50s g=%s%s" % (s, e, e - s, r["Queue_Id"], nm(r)[:50], r["Grid_Size_X"], ("   <-- idle %.1f" % gap) if gap > 3 else ""))
print("# prefill until the first decode kernel: %.1f us; whole step: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, (int(rows[end_step]["End_Timestamp"]) - t0) / 1e3))
