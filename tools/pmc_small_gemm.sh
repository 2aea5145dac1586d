# PMC counters of the small (32x32-tile) GEMM kernel, from the probe binary: where do its waves wait?
set -e
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_small_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o run -- $GRAFT_REPO_ROOT/tools/probes/probe_ops 3 > $out.log 2>&1 || { tail -3 $out.log; continue; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out "gemm_kernel<2, 2, 1, 1, false, false" | grep " 102400 " | head -1
done
