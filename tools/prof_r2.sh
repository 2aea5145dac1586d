# usage (on the GPU box, from the repo root): bash tools/prof_r2.sh <mode> <tag> [extra bench args]
# 1) bench.py normally -> gpurun_out/<tag>_bench_<mode>.json (with roofline.by_kernel + cpu_baseline)
# 2) the same command under rocprofv3 --kernel-trace --stats (no eager profile pass, no CPU leg)
#    -> gpurun_out/<tag>_<mode>_kernel_stats.txt
set -e
mode=$1; tag=$2; shift 2
root=$GRAFT_REPO_ROOT
python3 $root/bench.py --mode $mode --no-modes "$@" > $root/gpurun_out/${tag}_bench_${mode}.json 2> $root/gpurun_out/${tag}_bench_${mode}.err || { tail -20 $root/gpurun_out/${tag}_bench_${mode}.err; exit 1; }
cut -c1-400 $root/gpurun_out/${tag}_bench_${mode}.json
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_${tag}_${mode}
rm -rf $out
steps=10; warm=3
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o runc -- python3 $root/bench.py --mode $mode --steps $steps --warmup $warm --no-cpu-baseline --no-profile --no-modes --min-seconds 0 "$@" > $out.log 2>&1 || { tail -20 $out.log; exit 1; }
python3 $root/tools/profile_summary.py $out $((steps + warm)) > $root/gpurun_out/${tag}_${mode}_kernel_stats.txt
head -24 $root/gpurun_out/${tag}_${mode}_kernel_stats.txt | cut -c1-170
