#!/bin/bash
# MFMA utilisation of every kernel of the captured cfg2 train step (north_star: "MFMA utilisation against peak"):
# SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES per kernel, with the MFMA instruction count, from a rocprofv3 --pmc pass of
# `bench.py --mode train` (counters in their own run, with --kernel-trace only).   bash tools/pmc_mfma_train.sh <tag>
tag=${1:-r05_z}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmc_${tag}_mfma
rm -rf $out
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES --kernel-trace --output-format csv -d $out -o run -- python3 $root/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-modes --min-seconds 0 > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - $out > $root/gpurun_out/${tag}_pmc_mfma_train.txt <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    key = (r["Kernel_Name"], r["Dispatch_Id"])
    a = agg.setdefault(key, {"_t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
per = collections.OrderedDict()
for (k, _), c in agg.items():
    p = per.setdefault(k, collections.Counter())
    p["n"] += 1
    for n, v in c.items():
        p[n] += v
print("# MFMA-busy share of the busy SQ cycles per kernel of the captured cfg2 train step (all dispatches of the run, incl. the eager")
print("# warm-up pass): SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES; MOPS = MFMA operations in units of 512 (f32) / ... as the counter reports")
print("%-78s %6s %9s %9s %12s %12s" % ("kernel", "calls", "avg_us", "mfma_busy", "MOPS_F32", "MOPS_BF16"))
rows = []
for k, p in per.items():
    busy = p.get("SQ_BUSY_CYCLES", 0.0)
    name = k[k.find("::", 5) + 2:] if "::" in k else k
    rows.append((p["_t"], name[:78], p["n"], p["_t"] / p["n"], (p.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / busy) if busy else 0.0,
                 p.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / p["n"], p.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) / p["n"]))
for t, name, n, avg, share, f32, bf in sorted(rows, reverse=True)[:24]:
    print("%-78s %6d %9.1f %9.3f %12.4g %12.4g" % (name, n, avg, share, f32, bf))
PY
head -30 $root/gpurun_out/${tag}_pmc_mfma_train.txt | cut -c1-140
