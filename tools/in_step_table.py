#!/usr/bin/env python3
"""Kernel time per class INSIDE the captured step (two streams, kernels beside each other) from a rocprofv3
--kernel-trace run of `bench.py --mode <m>`: {class: {"us_per_step", "launches_per_step"}} for bench.py's
roofline.frac_in_step.   usage: in_step_table.py <rocprof dir> <steps run under the profiler> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_classes import CLASSES, classify  # noqa: E402


def main(d, steps, out):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    agg = collections.OrderedDict()
    other = 0.0
    once = collections.Counter()                     # kernels the step launches exactly once: their count IS the step count
    for r in csv.DictReader(open(f)):
        if r.get("Grid_Size_X"):
            grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
        else:
            grid = r.get("Grid_Size")
        key = classify(r["Kernel_Name"], grid)
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if key in ("adam", "vocab_ps"):
            once[key] += 1
        if key is None:
            other += us
            continue
        a = agg.setdefault(CLASSES[key], [0, 0.0])
        a[0] += 1
        a[1] += us
    # the trace also holds the eager pass the graph is captured after: divide by the passes actually traced
    steps = once.get("adam") or once.get("vocab_ps") or steps
    table = {"_source": "rocprofv3 --kernel-trace of the captured step (%s, %d passes traced incl. warm-up): kernel durations while "
                        "the step's two streams run beside each other; tools/in_step_table.py" % (os.path.basename(d.rstrip("/")), steps),
             "_unclassified_us_per_step": other / steps}
    for k, (n, us) in agg.items():
        table[k] = {"us_per_step": us / steps, "launches_per_step": n / steps, "avg_us": us / n}
    json.dump(table, open(out, "w"), indent=0)
    for k, v in table.items():
        print(k, v)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
