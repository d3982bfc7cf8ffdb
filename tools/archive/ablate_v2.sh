#!/bin/bash
# Ablation of conv3d_k3_v2 layers on the GPU box: AMX_DBG bit 1 = no DMA after the first stage, 2 = no MFMA, 4 = no output stores.
# usage: tools/ablate_v2.sh > gpurun_out/ablate.log
for shape in "32 64 32 64 4" "64 128 64 32 4" "64 0 64 32 4" "128 256 128 16 4" "128 0 128 16 4" "256 0 256 8 4" "32 0 64 32 4"; do
  for d in 0 1 2 4 3 5 6; do
    AMX_DBG=$d timeout 120 python tools/one_layer.py $shape 2>/dev/null | tail -1
  done
done
