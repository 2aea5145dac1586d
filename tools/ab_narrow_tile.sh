#!/bin/bash
for mode in forward train; do
  for rep in 1 2; do
    for kv in "" "ICK_PS_TILE_NARROW=10" "ICK_PS_TILE_NARROW=9"; do
      echo -n "$mode [$kv] : "
      env $kv python bench.py --mode $mode --no-modes --no-cpu-baseline --no-profile --min-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
    done
  done
done
