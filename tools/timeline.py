#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: one line per dispatch of the last full
train step (between two adam_clamp launches), with queue ids and idle gaps.  usage: timeline.py <dir> [marker]"""
import collections
import csv
import glob
import re
import sys


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "adam_clamp"
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))

    def nm(r):
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
        return m.group(1) if m else r["Kernel_Name"][:40]

    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    a, b = idx[-3], idx[-2]
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    print("# %d kernels, span %.1f us, queues %s" % (len(seg), (int(seg[-1]["End_Timestamp"]) - t0) / 1e3,
                                                    dict(collections.Counter(r["Queue_Id"] for r in seg))))
    busy_end = 0.0
    for r in seg:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        gap = s - busy_end if s > busy_end else 0.0
        busy_end = max(busy_end, e)
        print("%9.1f %9.1f %7.1f q%s %-48s g=%s,%s,%s%s" % (s, e, e - s, r["Queue_Id"], nm(r)[:48], r["Grid_Size_X"],
              r["Grid_Size_Y"], r["Grid_Size_Z"], ("   <-- idle %.1f" % gap) if gap > 3 else ""))


if __name__ == "__main__":
    main()
