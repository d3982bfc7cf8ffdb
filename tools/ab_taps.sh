#!/bin/bash
# contrastive step: sampled-tap route against the dense-tap route (AMX_DENSE_TAPS=1) on ONE box, graph replay
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  echo "sampled $(run --workload step --steps 20 --warmup 2)"
  echo "dense   $(AMX_EXPERIMENT=1 AMX_DENSE_TAPS=1 run --workload step --steps 20 --warmup 2)"
done
