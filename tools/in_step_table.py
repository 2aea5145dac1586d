#!/usr/bin/env python3
"""Kernel time per class INSIDE the captured step (two streams, kernels beside each other) from a rocprofv3
--kernel-trace run of `bench.py --mode <m> --config <c>`: {class: {"us_per_step", "launches_per_step", "avg_us"}} for
bench.py's roofline.frac_in_step.  The classes are found through the bench line of the same command (its by_kernel rows
carry the grids of their launches: tools/kernel_classes.py); the table records the build id it was made with
(ick_amd.build.source_id), and bench.py marks it stale when the tree has changed since.
usage: in_step_table.py <rocprof dir> <out.json> <bench line file> <marker kernel: launched once per step>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_classes import classify, load_keymap  # noqa: E402


def main(d, out, bench_json, marker):
    import ick_amd.build as build
    keymap = load_keymap(bench_json)
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    agg = collections.OrderedDict()
    other = 0.0
    steps = 0
    for r in csv.DictReader(open(f)):
        if r.get("Grid_Size_X"):
            grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
        else:
            grid = r.get("Grid_Size")
        if marker in r["Kernel_Name"]:
            steps += 1
        key = classify(r["Kernel_Name"], grid, keymap)
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if key is None:
            other += us
            continue
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += us
    assert steps > 0, "marker kernel %r not in the trace" % marker
    table = {"_source": "rocprofv3 --kernel-trace of the captured step (%s, %d passes traced incl. warm-up and capture): kernel "
                        "durations while the step's two streams run beside each other; tools/in_step_table.py" %
                        (os.path.basename(d.rstrip("/")), steps),
             "_build_id": build.source_id(), "_unclassified_us_per_step": other / steps}
    for k, (n, us) in agg.items():
        table[k] = {"us_per_step": us / steps, "launches_per_step": n / steps, "avg_us": us / n}
    json.dump(table, open(out, "w"), indent=0)
    for k, v in table.items():
        print(k, v)


if __name__ == "__main__":
    main(*sys.argv[1:5])
