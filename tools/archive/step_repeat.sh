#!/bin/bash
# the replayed step, N fresh processes on one box: how wide is the run-to-run spread?  usage: tools/step_repeat.sh [N]
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq 1 ${1:-8}); do
  python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 --workload step --steps 40 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done | tr '\n' ' '; echo
