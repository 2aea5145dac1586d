#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_round5_gpu.py tests/test_bench_sizes_gpu.py tests/test_training_gpu.py tests/test_adam_derive_gpu.py -x -q -m gpu > gpurun_out/r5_c5_tests.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r5_c5_tests.log
timeout -k 10 300 bash tools/ab_multi.sh - ICK_NO_LAYER_WGRAD_PS=1 > gpurun_out/r5_c5_ab_wgrad_ps.txt 2>&1
cat gpurun_out/r5_c5_ab_wgrad_ps.txt
timeout -k 10 200 bash tools/ab_env_list.sh "greedy" - > gpurun_out/r5_c5_greedy.txt 2>&1; cat gpurun_out/r5_c5_greedy.txt
