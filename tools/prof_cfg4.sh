set -e
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg4_train
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg4 --mode train --steps 8 --warmup 3 --no-cpu-baseline --no-modes --min-seconds 0 > $out.log 2>&1
tail -1 $out.log | cut -c1-100
