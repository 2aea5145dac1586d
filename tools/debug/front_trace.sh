#!/bin/bash
# kernel-level timeline of one captured train step (rocprofv3 --kernel-trace): what do the context-chain kernels do
# while conv1 / the image K/V projection run?
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_front
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_front -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-profile --no-cpu-baseline --no-modes --min-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_front.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_front.log | cut -c1-160
python3 $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/gpurun_out/prof_front > $GRAFT_REPO_ROOT/gpurun_out/front_timeline.txt
head -80 $GRAFT_REPO_ROOT/gpurun_out/front_timeline.txt
