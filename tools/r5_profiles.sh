#!/bin/bash
# Round-5 measurement artifacts (GPU box).  Per workload: the bench line (with roofline.by_kernel + its trace keys), rocprofv3
# kernel stats of the same command, the in-step table and the PMC traffic table joined through that line -- for ALL FIVE
# workloads the default bench line reports (train cfg2, forward cfg2, greedy cfg5, beam cfg5, train cfg4), so that no `modes`
# entry carries a null frac_in_step / traffic.  Then the device-timestamp timeline of the captured train step.
#   bash tools/r5_profiles.sh <tag> [workloads: "train:cfg2 forward:cfg2 ..."]
tag=${1:-r05_a}
todo=${2:-"train:cfg2 forward:cfg2 greedy:cfg5 beam:cfg5 train:cfg4"}
cd $GRAFT_REPO_ROOT
for spec in $todo; do
  mode=${spec%%:*}; cfg=${spec##*:}
  t=${tag}_${cfg}
  marker=packed_ce_rows_kernel      # once per pass, eager warm-up passes included (the optimizer kernel is not: flushes, no-op launches)
  [ $mode = forward ] && marker=caption_embed_kernel
  [ $mode = greedy ] && marker=dec_init
  [ $mode = beam ] && marker=dec_init
  timeout -k 10 400 bash tools/prof_r2.sh $mode $t --config $cfg --no-cpu-baseline > gpurun_out/${t}_${mode}.log 2>&1 || { tail -5 gpurun_out/${t}_${mode}.log; exit 1; }
  tail -2 gpurun_out/${t}_${mode}.log | cut -c1-150
  python3 tools/in_step_table.py gpurun_out/prof_${t}_${mode} gpurun_out/r05_in_step_${mode}_${cfg}.json gpurun_out/${t}_bench_${mode}.json $marker > gpurun_out/${t}_in_step_${mode}.txt 2>&1 || { tail -3 gpurun_out/${t}_in_step_${mode}.txt; exit 1; }
  timeout -k 10 500 bash tools/pmc_traffic.sh $mode $t --config $cfg > gpurun_out/${t}_pmc_${mode}.log 2>&1 || { tail -5 gpurun_out/${t}_pmc_${mode}.log; exit 1; }
  python3 tools/traffic_table.py gpurun_out/r05_traffic_${mode}_${cfg}.json gpurun_out/${t}_bench_${mode}.json gpurun_out/pmc_${t}_${mode}_FETCH_SIZE gpurun_out/pmc_${t}_${mode}_WRITE_SIZE > gpurun_out/${t}_traffic_${mode}.txt 2>&1 || { tail -3 gpurun_out/${t}_traffic_${mode}.txt; exit 1; }
  head -6 gpurun_out/${t}_traffic_${mode}.txt | cut -c1-150
  rm -rf gpurun_out/pmc_${t}_${mode}_FETCH_SIZE gpurun_out/pmc_${t}_${mode}_WRITE_SIZE      # large CSVs: the tables are what is kept
done
python3 tools/timeline.py gpurun_out/prof_${tag}_cfg2_train packed_ce_reduce > gpurun_out/${tag}_train_kernel_timeline.txt 2>&1
ICK_TIMESTAMPS=1 timeout -k 10 200 python3 tools/host_bound.py > gpurun_out/${tag}_train_device_timestamps.txt 2>&1; tail -42 gpurun_out/${tag}_train_device_timestamps.txt
