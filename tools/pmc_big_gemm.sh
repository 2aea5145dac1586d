# PMC counters of the large GEMM kernels (feature projection, cross K/V, vocabulary, 4096^3 reference shape) from the
# probe binary: MFMA busy, wave stalls, LDS conflicts.  Separate passes per counter set (8 SQ slots).
set -e
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_big_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o run -- $GRAFT_REPO_ROOT/tools/probes/probe_ops 2 > $out.log 2>&1 || { tail -3 $out.log; continue; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out "gemm_kernel<" | grep -E "grid +(250880|1455104|803840|1048576) " | cut -c1-400
done
