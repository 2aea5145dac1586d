#!/bin/bash
# ceiling experiments: which classes of launches is the captured train step's time sensitive to? (results are wrong with
# ICK_EXP set; only the timing of the remaining kernels is meaningful)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 bash tools/ab_multi.sh - ICK_EXP=wgrad=small ICK_EXP=wgrad=all ICK_EXP=attn ICK_EXP=pack ICK_EXP=attn,wgrad=all,pack > gpurun_out/r5_c2_ceilings.txt 2>&1
cat gpurun_out/r5_c2_ceilings.txt
timeout -k 10 300 bash tools/ab_env_list.sh "forward" - ICK_EXP=attn ICK_EXP=pack > gpurun_out/r5_c2_ceilings_fwd.txt 2>&1
cat gpurun_out/r5_c2_ceilings_fwd.txt
