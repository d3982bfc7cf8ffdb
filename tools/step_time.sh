#!/bin/bash
# contrastive step time (graph replay), a few repetitions on one box
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
  python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 --workload step --steps 20 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['value'], d['ms_per_step'])"
done
python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_step'])"
