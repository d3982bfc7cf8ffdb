#!/bin/bash
# ablations of the stem-fed z-march launch on the experiment library: AMX_DBG 2 = no consumer sweep, 4 = no stores,
# 64 = stem waves only publish, 128 = converter only waits for its DMA and publishes
export AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so LP_IGNORE_OVERFLOW=1
for dbg in ${@:-0 6 70 134 198}; do
  echo "== AMX_DBG=$dbg"; AMX_DBG=$dbg timeout 200 python tools/layer_profile.py anatomix 4 2>&1 | grep -E "m 0|m 3 "
done
