#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
for d in 0 2 4 6; do AMX_DBG=$d timeout 120 python tools/one_layer.py 32 0 32 64 4 2>/dev/null | tail -1; done
AMX_TRACE=1 timeout 120 python tools/one_layer.py 32 0 32 64 4 2>&1 | grep -i "trace" | head -4 | cut -c1-330
