# PMC counters of the pre-split GEMM kernel (csrc/gemm_ps.hip) on two shapes: wave cycles / stalls, MFMA busy, LDS
# conflicts, instruction counts.  Separate passes per counter set (MI355X_MICROARCH.md).   bash tools/pmc_gemm_ps.sh [shape ...]
cd /tmp && export TMPDIR=/tmp
shapes=${@:-cross_kv conv1}
for shape in $shapes; do
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_ps_${shape}_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/tools/gemm_ps_bench.py --child $shape > $out.log 2>&1 || { tail -3 $out.log; continue; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out "gemm_ps_kernel" | cut -c1-420
done
done
