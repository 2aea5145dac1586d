set -e
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_train2
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train2 -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes --min-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_train2.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_train2.log | cut -c1-200
