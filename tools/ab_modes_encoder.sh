#!/bin/bash
# Encoder.conv1 inside the captured graph (attach_encoder) against the stand-alone Encoder launch in front of it, per
# bench mode, interleaved on one box.   usage: bash tools/ab_modes_encoder.sh [modes...]
for mode in ${@:-forward greedy beam}; do
  for rep in 1 2; do
    for kv in "" "ICK_BENCH_SEPARATE_ENCODER=1"; do
      echo -n "$mode [$kv] : "
      env $kv python bench.py --mode $mode --no-modes --no-cpu-baseline --no-profile --min-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
    done
  done
done
