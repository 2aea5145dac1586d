#!/bin/bash
# round-end artifacts: bench JSON + rocprofv3 kernel stats of every mode (tools/prof_r2.sh), default bench line
# usage (GPU box): bash tools/final_profiles.sh <tag>
tag=${1:-r03_i}
cd $GRAFT_REPO_ROOT
for spec in "train" "forward" "greedy --config cfg5" "beam --config cfg5" "train --config cfg4"; do
  set -- $spec
  mode=$1; shift
  t=$tag; [ "$1" = "--config" ] && [ "$2" = "cfg4" ] && t=${tag}_cfg4
  timeout -k 10 400 bash tools/prof_r2.sh $mode $t "$@" > gpurun_out/${t}_${mode}.log 2>&1 || { tail -5 gpurun_out/${t}_${mode}.log; exit 1; }
  tail -3 gpurun_out/${t}_${mode}.log | cut -c1-150
done
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err && cut -c1-300 gpurun_out/${tag}_bench_default.json
