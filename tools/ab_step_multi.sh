#!/bin/bash
# several experiment settings of the contrastive step on one box, alternating: tools/ab_step_multi.sh "A=1" "B=1 C=2" ...  ("-" = defaults)
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 --workload step --steps 20 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then echo "default: $(run)"; else echo "$kv: $(env AMX_EXPERIMENT=1 $kv bash -c "$(declare -f run); run")"; fi
  done
done
