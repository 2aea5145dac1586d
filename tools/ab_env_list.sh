#!/bin/bash
# Interleaved A/B of environment variants on given bench modes, one box.   usage: bash tools/ab_env_list.sh "forward train" "A=1" "A=2 B=3" ...  ("-" = no variable)
modes=$1; shift
for mode in $modes; do
  for rep in 1 2; do
    for kv in "$@"; do
      [ "$kv" = "-" ] && kv=""
      echo -n "$mode [$kv] : "
      env $kv python bench.py --mode $mode --no-modes --no-cpu-baseline --no-profile --min-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
    done
  done
done
