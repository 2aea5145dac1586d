#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_adam_derive_gpu.py -x -q -m gpu 2>&1 | tail -3
ICK_TIMESTAMPS=1 timeout -k 10 200 python tools/host_bound.py > gpurun_out/r5_c4_stamps.txt 2>&1; tail -45 gpurun_out/r5_c4_stamps.txt
timeout -k 10 400 bash tools/prof_r2.sh train r5_c4 --no-cpu-baseline --no-profile > gpurun_out/r5_c4_prof.log 2>&1; tail -30 gpurun_out/r5_c4_prof.log | cut -c1-150
