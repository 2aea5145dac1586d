#!/bin/bash
# A/B of a build-time switch on one box: default library vs a variant built with extra hipcc flags, interleaved.
# usage: bash tools/ab_build.sh <tag> "<flags>" <mode>
tag=$1; flags=$2; mode=${3:-train}
root=$GRAFT_REPO_ROOT
python3 - <<PY
import os, subprocess, sys
sys.path.insert(0, "$root")
import ick_amd.build as b
out = os.path.join("$root", "gpurun_out", "lib_$tag.so")
subprocess.check_call([b.HIPCC] + b.FLAGS + "$flags".split() + ["-shared", "-o", out] + b.sources())
PY
for rep in 1 2; do
  for lib in "" "$root/gpurun_out/lib_$tag.so"; do
    echo -n "$mode [${lib:+$tag}] : "
    ICK_LIB_PATH=$lib python3 $root/bench.py --mode $mode --no-modes --no-cpu-baseline --no-profile --min-seconds 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
done
