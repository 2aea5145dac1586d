#!/bin/bash
# train-step time with the large-tile GEMMs declaring more LDS than they use (fewer workgroups per CU)
out=gpurun_out/occupancy_sweep.txt
: > $out
for cfg in "0 0" "45000 0" "56000 0" "0 45000" "0 56000" "56000 56000" "45000 45000" "56000 45000"; do
  set -- $cfg
  echo "== ICK_GEMM_LDS_SINGLE=$1 ICK_GEMM_LDS_GROUP=$2" >> $out
  ICK_GEMM_LDS_SINGLE=$1 ICK_GEMM_LDS_GROUP=$2 timeout -k 10 120 python tools/host_bound.py 2>&1 | grep "host issue" >> $out || exit 1
done
for cfg in "56000 0" "0 56000"; do
  set -- $cfg
  echo "== timeline ICK_GEMM_LDS_SINGLE=$1 ICK_GEMM_LDS_GROUP=$2" >> $out
  ICK_TIMESTAMPS=1 ICK_GEMM_LDS_SINGLE=$1 ICK_GEMM_LDS_GROUP=$2 timeout -k 10 120 python tools/host_bound.py 2>&1 | grep " us  " >> $out || exit 1
done
