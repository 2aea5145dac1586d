#!/bin/bash
# Several environment variants of the default train bench on one box, interleaved twice (devices of the pool differ by a
# few per cent, so only runs of one call compare).   usage: bash tools/ab_multi.sh "A=1" "B=2 C=3" ... ("-" = default)
for rep in 1 2; do
  for kv in "$@"; do
    [ "$kv" = "-" ] && kv=""
    echo -n "[$kv] : "
    env $kv python bench.py --no-modes --no-cpu-baseline --no-profile --min-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
done
