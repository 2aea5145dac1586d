python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for p in 0 1 -1; do echo "ICK_BWD_SIDE_PRIO=$p"; ICK_BWD_SIDE_PRIO=$p python bench.py --no-modes --no-cpu-baseline --no-profile --min-seconds 1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
