#!/bin/bash
# front of the train step (context chain beside conv1 + K/V projection): serial vs overlapped vs occupancy-capped GEMMs
out=gpurun_out/front_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" ICK_TIMESTAMPS=1 timeout -k 10 120 python tools/host_bound.py 2>&1 | grep -E "host issue|context chain|K/V projection done|reaches cross|decoder layers done|A: end|B: end" >> $out || exit 1; }
run ICK_X=0
run ICK_NO_FWD_OVERLAP=1
run ICK_GEMM_LDS_SINGLE=60000
run ICK_GEMM_LDS_SINGLE=84000
run ICK_NO_ROWCHAIN=1
