# Where do the waves of the decode kernels spend their time?  SQ counters (several passes) + L2 hit / miss counters of one
# greedy bench run, averaged per kernel.   usage (GPU box, repo root): bash tools/pmc_decode.sh <tag> [mode]
set -e
tag=$1; mode=${2:-greedy}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_${tag}_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o run -- python3 $root/bench.py --mode $mode --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-modes --min-seconds 0 > $out.log 2>&1 || { tail -5 $out.log; continue; }
done
python3 - <<PY > $root/gpurun_out/${tag}_pmc_decode.txt
import collections, csv, glob
agg = collections.OrderedDict()
for i in range(1, $i + 1):
    fs = glob.glob("$root/gpurun_out/pmc_${tag}_%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "dec_" not in k:
            continue
        k = k[k.find("dec_"):][:40]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, c in per.items():
        a = agg.setdefault(k, collections.OrderedDict())
        for n, v in c.items():
            a[n] = v / len(disp[k])
for k, c in agg.items():
    print(k)
    for n, v in c.items():
        print("    %-34s %14.1f" % (n, v))
PY
cat $root/gpurun_out/${tag}_pmc_decode.txt
