#!/bin/bash
# round-end validation: full GPU suite, smoke, default bench (artifact of the round)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/final/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1 && tail -2 gpurun_out/final/smoke.log &&
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err && cut -c1-600 gpurun_out/final/bench_default.json
