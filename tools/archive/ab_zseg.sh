#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
run() { python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  unset AMX_OLD_ZSEG; echo "new step $(run --workload step --steps 20 --warmup 2)"
  export AMX_OLD_ZSEG=1; echo "old step $(run --workload step --steps 20 --warmup 2)"
done
