#!/bin/bash
# z-march ablation on the 64^3 layers (experiment library): AMX_DBG 1 no DMA after the first planes, 2 no MFMA sweep, 4 no stores.
cd ${GRAFT_REPO_ROOT:-.}
export AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
for shape in "32 0 32 64 4" "16 0 32 64 4" "16 0 16 128 4"; do
  for d in 0 1 2 4 3 6 7; do
    AMX_DBG=$d timeout 120 python tools/one_layer.py $shape 2>/dev/null | tail -1
  done
  AMX_TRACE=1 timeout 120 python tools/one_layer.py $shape 2>&1 | grep -i "trace" | head -6 | cut -c1-400
done
