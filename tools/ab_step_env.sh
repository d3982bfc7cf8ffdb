#!/bin/bash
# A/B of a Python-level experiment switch on the contrastive step (one box, alternating): tools/ab_step_env.sh VAR=VALUE [reps]
cd ${GRAFT_REPO_ROOT:-.}
KV=$1; REPS=${2:-3}
run() { python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 --workload step --steps 20 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in $(seq $REPS); do
  echo "default $(run)"
  echo "$KV $(env AMX_EXPERIMENT=1 $KV bash -c "$(declare -f run); run")"
done
