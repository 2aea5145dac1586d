#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_adam_derive_gpu.py tests/test_rowchain_gpu.py tests/test_training_gpu.py tests/test_round5_gpu.py -x -q -m gpu > gpurun_out/r5_c3_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r5_c3_tests.log
timeout -k 10 300 bash tools/ab_multi.sh - ICK_ADAM_DERIVE=0 > gpurun_out/r5_c3_ab_derive.txt 2>&1
cat gpurun_out/r5_c3_ab_derive.txt
