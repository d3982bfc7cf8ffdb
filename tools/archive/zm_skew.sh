#!/bin/bash
# z-march: phase skew between the two consumer waves of a SIMD (experiment library, AMX_DBG bits 16..23 = n x 512 cycles)
cd ${GRAFT_REPO_ROOT:-.}
export AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
for shape in "32 0 32 64 4" "16 0 16 128 4"; do
  for n in 0 2 3 4 5 6 8; do
    AMX_DBG=$((n << 16)) timeout 120 python tools/one_layer.py $shape 2>/dev/null | tail -1
  done
done
AMX_DBG=$((5 << 16)) AMX_TRACE=1 timeout 120 python tools/one_layer.py 32 0 32 64 4 2>&1 | grep -i "trace" | head -2 | cut -c1-330
