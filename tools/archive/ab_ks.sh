#!/bin/bash
# A/B on ONE box: conv3d_k3_ks on / off (experiment library, AMX_NO_KS=1) for the headline forward and the contrastive step
cd ${GRAFT_REPO_ROOT:-.}
export AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
run() { python bench.py --no-secondary --no-cpu-baseline --no-parity --sustain 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for off in 0 1; do
    if [ $off = 1 ]; then export AMX_NO_KS=1; else unset AMX_NO_KS; fi
    echo "no_ks=$off headline $(run --steps 60)"
    echo "no_ks=$off step     $(run --workload step --steps 20 --warmup 2)"
  done
done
