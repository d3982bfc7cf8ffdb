#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
for d in 9 10 12 14; do echo "AMX_DBG=$d"; AMX_DBG=$d timeout 120 python tools/one_layer.py 32 0 32 64 4 2>&1 | grep -i "trace" | head -2 | cut -c1-300; done
