#!/bin/bash
# A/B of one environment switch on the default train bench, alternating runs on the same box (devices of the pool differ
# by a few per cent).   usage: bash tools/ab_env.sh VAR=VALUE [bench args]
kv=$1; shift
for rep in 1 2; do
  echo -n "default        : "; python bench.py --no-modes --no-cpu-baseline --no-profile --min-seconds 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
  echo -n "$kv : "; env $kv python bench.py --no-modes --no-cpu-baseline --no-profile --min-seconds 1 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
