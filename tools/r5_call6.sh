#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 400 bash tools/prof_r2.sh train r5_c6 --no-cpu-baseline --no-profile > gpurun_out/r5_c6_prof.log 2>&1; tail -32 gpurun_out/r5_c6_prof.log | cut -c1-150
