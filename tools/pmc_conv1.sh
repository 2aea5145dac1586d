# HBM traffic of the dominant kernel (Encoder.conv1 GEMM) from the PMC counters: separate passes for FETCH_SIZE and
# WRITE_SIZE (MI355X_MICROARCH.md: they do not fit one pass; FETCH_SIZE under-reports wide streaming reads by 2x)
set -e
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  rm -rf $out
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/bench.py --mode forward --steps 2 --warmup 1 --no-cpu-baseline --no-modes --min-seconds 0 > $out.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out gemm_kernel | grep "250880" | head -2
done
