#!/bin/bash
# conv3d_k3_ks under the experiment library: phase traces (s_memtime stamps of wave 0) and ablations of the deep-level shapes at batch 4.
cd ${GRAFT_REPO_ROOT:-.}
export AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
for shape in "64 0 64 32 4" "128 0 128 16 4" "256 0 256 8 4"; do
  echo "== $shape"
  AMX_TRACE=1 timeout 120 python tools/one_layer.py $shape 2>&1 | grep -E "trace|us " | cut -c1-900
  for dbg in 0 32 64 1 2 16 18 19; do
    AMX_DBG=$dbg timeout 120 python tools/one_layer.py $shape 2>&1 | tail -1
  done
done
