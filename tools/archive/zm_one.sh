#!/bin/bash
# z-march layers, product library (new) against lib_old.bin.  usage (GPU box): bash tools/zm_one.sh
cd ${GRAFT_REPO_ROOT:-.}
for shape in "16 0 16 128 4" "16 0 16 128 2" "16 0 16 64 4"; do
  echo "new: $(python tools/one_layer.py $shape 2>/dev/null | tail -1)"
  echo "old: $(AMX_LIB_PATH=$PWD/anatomix_amd/csrc/lib_old.bin python tools/one_layer.py $shape 2>/dev/null | tail -1)"
done
