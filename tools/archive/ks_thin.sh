#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export AMX_LIB_PATH=$PWD/anatomix_amd/csrc/libanatomix_amd_exp.so
for shape in "256 0 256 8 4" "128 0 256 8 4"; do
  for th in 0 1; do
    if [ $th = 1 ]; then export AMX_KS_THIN=1; else unset AMX_KS_THIN; fi
    echo "thin=$th $(timeout 120 python tools/one_layer.py $shape 2>&1 | tail -1)"
  done
done
