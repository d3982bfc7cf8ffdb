#!/bin/bash
# A/B of several library builds on ONE box (headline forward, alternating, two rounds): tools/ab_many.sh lib1.so lib2.so ...
# (paths relative to anatomix_amd/csrc/ab/)
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 --steps 60 "${EXTRA[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
EXTRA=()
for rep in 1 2 3; do
  for l in "$@"; do
    echo "$l $(AMX_LIB_PATH=$PWD/anatomix_amd/csrc/ab/$l run)"
  done
done
