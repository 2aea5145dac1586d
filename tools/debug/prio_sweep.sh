#!/bin/bash
# wave priority of the backward chain kernels vs the weight-gradient GEMMs beside them (compile-time switch: variant libraries)
cd $GRAFT_REPO_ROOT
for p in 3 2 1 0; do
  python - $p <<PY
import subprocess, sys, os
sys.path.insert(0, os.getcwd())
import ick_amd.build as b
out = "gpurun_out/libick_prio%s.so" % sys.argv[1]
subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DICK_CHAIN_PRIO_BWD=" + sys.argv[1], "-shared", "-o", out] + b.sources())
PY
  echo "bwd priority $p"
  for i in 1 2; do ICK_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_out/libick_prio$p.so python bench.py --no-profile --no-cpu-baseline 2>/dev/null | cut -c1-130; done
done
