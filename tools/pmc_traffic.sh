# HBM traffic per launch of every kernel of one bench mode, from the PMC counters: separate rocprofv3 passes for
# FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md: they do not fit one pass; FETCH_SIZE is in 64-byte units reported as
# KB and counts wide streaming reads at half their bytes on gfx950 -> doubled in the summary).
# usage (GPU box, repo root): bash tools/pmc_traffic.sh <mode> <tag> [extra bench args]   ->  gpurun_out/<tag>_pmc_<mode>.txt
set -e
mode=$1; tag=$2; shift 2
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_${tag}_${mode}_$c
  rm -rf $out
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o run -- python3 $root/bench.py --mode $mode --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-modes --min-seconds 0 "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
done
python3 $root/tools/pmc_traffic_summary.py $root/gpurun_out/pmc_${tag}_${mode}_FETCH_SIZE $root/gpurun_out/pmc_${tag}_${mode}_WRITE_SIZE > $root/gpurun_out/${tag}_pmc_${mode}.txt
head -30 $root/gpurun_out/${tag}_pmc_${mode}.txt | cut -c1-200
