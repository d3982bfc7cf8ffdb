#!/bin/bash
# A/B of two library builds on ONE box, contrastive step: anatomix_amd/csrc/libanatomix_amd.so (new) against
# anatomix_amd/csrc/lib_old.bin (built from another revision and copied there by hand; not tracked).  usage (GPU box): bash tools/ab_libs_step.sh
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 --workload step --steps 20 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  echo "new $(run)"
  echo "old $(AMX_LIB_PATH=$PWD/anatomix_amd/csrc/lib_old.bin run)"
done
