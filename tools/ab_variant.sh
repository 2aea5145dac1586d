# usage (GPU box, repo root): bash tools/ab_variant.sh <tag> "<extra hipcc flags>" <bench args...>
# builds a library variant with extra compile-time switches into gpurun_out/lib_<tag>.so and runs bench.py on it
set -e
tag=$1; flags=$2; shift 2
root=$GRAFT_REPO_ROOT
python3 - <<PY
import os, subprocess, sys
sys.path.insert(0, "$root")
import ick_amd.build as b
out = os.path.join("$root", "gpurun_out", "lib_$tag.so")
subprocess.check_call([b.HIPCC] + b.FLAGS + "$flags".split() + ["-shared", "-o", out] + b.sources())
PY
ICK_LIB_PATH=$root/gpurun_out/lib_$tag.so python3 $root/bench.py "$@" --no-cpu-baseline --no-profile --no-modes | cut -c1-160
