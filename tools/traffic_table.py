#!/usr/bin/env python3
"""HBM bytes per launch and kernel class from the PMC passes of tools/pmc_traffic.sh (FETCH_SIZE, WRITE_SIZE directories of
one bench mode), joined onto bench.py's class names through the bench line of the same command (tools/kernel_classes.py).
Classes made of several kernels get the launch-weighted mean, the decode step the SUM over one token's launches.  The table
records the build id it was made with.   usage: traffic_table.py <out.json> <bench line file> <fetch dir> <write dir>"""
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

DECODE = "fused decode step (3 kernels / layer + head + vocabulary)"


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        key = (r["Kernel_Name"], r["Grid_Size"], r["Dispatch_Id"])
        per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
    return per


def main(out, bench_json, fetch_dir, write_dir):
    import ick_amd.build as build
    keymap = load_keymap(bench_json)
    table = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_traffic.sh -> "
                        "tools/traffic_table.py), FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide streaming "
                        "reads; bytes per launch (decode step: per token = all its launches)",
             "_build_id": build.source_id()}
    rd, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    agg = collections.OrderedDict()
    tokens = 0                                    # decode steps = launches of the vocabulary kernel
    for (k, g, did), v in rd.items():
        key = classify(k, g, keymap)
        if key is None:
            continue
        if "dec_vocab_kernel" in k or "dec_headvocab_kernel" in k:
            tokens += 1
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += 2.0 * v * 1024
    for (k, g, did), v in wr.items():
        key = classify(k, g, keymap)
        if key in agg:
            agg[key][2] += v * 1024
    for key, (n, r, w) in agg.items():
        per = (r + w) / n
        if key == DECODE:
            per = (r + w) / max(tokens, 1)        # all launches of one token step
        table[key] = per
        print("%-100s launches %5d  %9.2f MB per %s" % (key[:100], n, per / 1e6, "token" if key == DECODE else "launch"))
    json.dump(table, open(out, "w"), indent=0)


if __name__ == "__main__":
    main(*sys.argv[1:5])
