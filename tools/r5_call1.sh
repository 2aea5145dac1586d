#!/bin/bash
# round 5, first GPU call: the new bench-path parity tests + A/B of existing knobs in the round-4 state
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_round5_gpu.py -x -q -m gpu > gpurun_out/r5_c1_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5_c1_tests.log
timeout -k 10 240 bash tools/ab_multi.sh - ICK_BWD_SUBGROUPS=2 ICK_BWD_SUBGROUPS=0 > gpurun_out/r5_c1_ab_subgroups.txt 2>&1
cat gpurun_out/r5_c1_ab_subgroups.txt
timeout -k 10 240 bash tools/ab_env_list.sh "greedy" - ICK_PS_NARROW_MIN=0 "ICK_PS_NARROW_MIN=0 ICK_PS_TILE_NARROW=8" > gpurun_out/r5_c1_ab_conv1_b32.txt 2>&1
cat gpurun_out/r5_c1_ab_conv1_b32.txt
