#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r5_full_tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -8 gpurun_out/r5_full_tests.log
exit $rc
