#!/bin/bash
# Round-4 measurement artifacts (GPU box): per mode bench JSON + rocprofv3 kernel stats, the in-step table of the train and
# forward steps, PMC traffic tables, device-timestamp timeline of the captured train step.   bash tools/round4_profiles.sh <tag>
tag=${1:-r04_x}
cd $GRAFT_REPO_ROOT
for spec in "train" "forward" "greedy --config cfg5" "beam --config cfg5" "train --config cfg4"; do
  set -- $spec
  mode=$1; shift
  t=$tag; [ "$1" = "--config" ] && [ "$2" = "cfg4" ] && t=${tag}_cfg4
  timeout -k 10 400 bash tools/prof_r2.sh $mode $t "$@" > gpurun_out/${t}_${mode}.log 2>&1 || { tail -5 gpurun_out/${t}_${mode}.log; exit 1; }
  tail -2 gpurun_out/${t}_${mode}.log | cut -c1-150
done
python3 tools/in_step_table.py gpurun_out/prof_${tag}_train 13 gpurun_out/${tag}_in_step_train.json > gpurun_out/${tag}_in_step_train.txt 2>&1 || tail -3 gpurun_out/${tag}_in_step_train.txt
python3 tools/in_step_table.py gpurun_out/prof_${tag}_forward 13 gpurun_out/${tag}_in_step_forward.json > gpurun_out/${tag}_in_step_forward.txt 2>&1 || tail -3 gpurun_out/${tag}_in_step_forward.txt
python3 tools/timeline.py gpurun_out/prof_${tag}_train > gpurun_out/${tag}_train_kernel_timeline.txt 2>&1
for mode in train forward greedy; do
  timeout -k 10 500 bash tools/pmc_traffic.sh $mode $tag > gpurun_out/${tag}_pmc_${mode}.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc_${mode}.log; }
done
python3 tools/traffic_table.py gpurun_out/${tag}_traffic.json gpurun_out/pmc_${tag}_train_FETCH_SIZE gpurun_out/pmc_${tag}_train_WRITE_SIZE gpurun_out/pmc_${tag}_greedy_FETCH_SIZE gpurun_out/pmc_${tag}_greedy_WRITE_SIZE > gpurun_out/${tag}_traffic.txt 2>&1; tail -20 gpurun_out/${tag}_traffic.txt | cut -c1-160
ICK_TIMESTAMPS=1 timeout -k 10 200 python3 tools/host_bound.py > gpurun_out/${tag}_train_device_timestamps.txt 2>&1; tail -45 gpurun_out/${tag}_train_device_timestamps.txt
