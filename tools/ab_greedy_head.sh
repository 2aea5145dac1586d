#!/bin/bash
for rep in 1 2; do
  for kv in "" "ICK_DECODE_SPLIT_HEAD=1"; do
    echo -n "[$kv] : "
    env $kv python bench.py --mode greedy --no-modes --no-cpu-baseline --no-profile --min-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  done
done
