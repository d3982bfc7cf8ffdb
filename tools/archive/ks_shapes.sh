#!/bin/bash
# Deep-level layer shapes of the 6 M network at batch 4: conv3d_k3_ks (product library) against conv3d_k3_v2 (the -DAMX_EXPERIMENT
# library with AMX_NO_KS=1) on ONE box.  usage (GPU box): bash tools/ks_shapes.sh
cd ${GRAFT_REPO_ROOT:-.}
X=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
for shape in "32 0 64 32 4" "64 0 64 32 4" "64 0 128 16 4" "128 0 128 16 4" "128 0 256 8 4" "256 0 256 8 4"; do
  a=$(timeout 120 python tools/one_layer.py $shape 2>&1 | tail -1)
  b=$(AMX_LIB_PATH=$X AMX_NO_KS=1 timeout 120 python tools/one_layer.py $shape 2>&1 | tail -1)
  echo "ks: $a"
  echo "v2: $b"
done
