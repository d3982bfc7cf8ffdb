#!/bin/bash
# experiment library: conv3d_k3_ks with two un-prefetched workgroups per CU (AMX_KS_PP=1) against the default configurations
cd ${GRAFT_REPO_ROOT:-.}
export AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
for shape in "32 0 64 32 4" "64 0 64 32 4" "64 0 128 16 4"; do
  for pp in 0 1; do
    echo "pp=$pp $(AMX_KS_PP=$pp timeout 120 python tools/one_layer.py $shape 2>&1 | tail -1)"
  done
done
AMX_KS_PP=1 timeout 300 python -m pytest tests/test_conv_ks_gpu.py -x -q 2>&1 | tail -3
