#!/bin/bash
# Ablations of the generic conv kernel on the deep-layer shapes (AMX_DBG: 1 no DMA after the first stage, 2 no MFMA sweep, 4 no stores)
for shape in "32 64 32 64 4" "64 0 64 32 4" "64 128 64 32 4" "128 0 128 16 4" "128 256 128 16 4" "256 0 256 8 4" "32 0 32 128 4" "32 64 32 128 4" "64 0 64 64 4"; do
  for dbg in 0 1 2 4 3; do
    AMX_DBG=$dbg python tools/one_layer.py $shape 2>&1 | tail -1
  done
done
