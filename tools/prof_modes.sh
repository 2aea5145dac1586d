# usage: bash tools/prof_modes.sh <tag>   -- rocprofv3 kernel traces of the train and forward benches (GPU box)
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
for mode in train forward; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_${mode}_$tag
  rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --mode $mode --steps 10 --warmup 3 --no-cpu-baseline --no-modes --min-seconds 0 > $out.log 2>&1
  tail -1 $out.log | cut -c1-120
done
