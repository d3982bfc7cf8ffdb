#!/bin/bash
# Times the conv3d_k3_v2 layer shapes of the 6 M network at batch 4 (AMX_DBG=0) -- quick A/B for kernel changes.
for shape in "32 64 32 64 4" "64 128 64 32 4" "64 0 64 32 4" "32 0 64 32 4" "128 256 128 16 4" "128 0 128 16 4" "64 0 128 16 4" "256 0 256 8 4" "128 0 256 8 4"; do
  timeout 120 python tools/one_layer.py $shape 2>/dev/null | tail -1
done
