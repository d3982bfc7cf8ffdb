#!/bin/bash
# like ab_env.sh, three alternations of 40 steps.  usage: tools/ab_env3.sh VAR=value
cd ${GRAFT_REPO_ROOT:-.}
KV=$1; shift
run() { python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 --workload step --steps 40 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "default  $(run "$@")"
  echo "$KV $(env $KV bash -c "$(declare -f run); run $*")"
done
