#!/bin/bash
# A/B of two library builds on ONE box, forward benches (headline + anatomix-dev): libanatomix_amd.so (new) against lib_old.bin.
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  echo "new headline $(run --steps 40)"
  echo "old headline $(AMX_LIB_PATH=$PWD/anatomix_amd/csrc/lib_old.bin run --steps 40)"
  echo "new dev      $(run --variant anatomix-dev --batch 4 --steps 8 --warmup 2)"
  echo "old dev      $(AMX_LIB_PATH=$PWD/anatomix_amd/csrc/lib_old.bin run --variant anatomix-dev --batch 4 --steps 8 --warmup 2)"
done
