#!/bin/bash
# per-kernel times of the ViT forward for the product-kernel launch variants (env AMX_GEMM_WS = weight-stationary kernel on / off,
# AMX_GEMM_WSMT = its row tiles per wave, AMX_GEMM_PF / AMX_GEMM_MT = prefetch distance / forced row tiles of the direct kernel)
for cfg in "0 4" "1 4" "1 2"; do
  set -- $cfg
  echo "=== AMX_GEMM_WS=$1 AMX_GEMM_WSMT=$2"
  AMX_GEMM_WS=$1 AMX_GEMM_WSMT=$2 tools/prof_fwd.sh r03vs --variant anatomix-dev-vit --warmup 3 | grep -E "gemm_kernel" | cut -c1-75,110-160
  grep -o '"value": [0-9.]*' gpurun_out/r03vs_prof.log | head -1
done
