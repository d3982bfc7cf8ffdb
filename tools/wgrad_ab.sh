#!/bin/bash
# A/B of the weight-gradient kernel's experiment switches (exp library): AMX_WGRAD_DBG 1 = no MFMA sweep, 2 = no DMA.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export AMX_EXPERIMENT=1 AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
for d in 0 1 2 3; do
  echo "== AMX_WGRAD_DBG=$d"
  AMX_WGRAD_DBG=$d WG_NOCHECK=1 timeout 200 python tools/wgrad_layers.py 2>&1 | grep -v amdgpu.ids
done
