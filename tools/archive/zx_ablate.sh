#!/bin/bash
# ablations of the fused 32 -> 32 kernel (AMX_ZX_DBG: 2 no MFMA sweeps, 32 no mx sweeps, 64 no main sweeps, 4 no stores, 16 converters
# idle after the first ring fill); ablated runs compute garbage, layer_profile.py swallows the overflow guard's reports
# the switches exist only in an ablation build of the kernel file (amx_conv3d_zx.hip, ZX_DBG); the production object is restored at the end
rm -f anatomix_amd/csrc/amx_conv3d_zx.o; make -s -C anatomix_amd/csrc EXTRA=-DAMX_ZX_ABLATE > /dev/null 2>&1
for dbg in ${DBGS:-0 2 4 16 32 64 18}; do
  AMX_ZX_DBG=$dbg LP_IGNORE_OVERFLOW=1 timeout 100 python tools/layer_profile.py anatomix-dev 4 2>&1 | grep -E "m 3 |Error|error" | sed "s/^/dbg=$dbg /" | cut -c1-130
done
rm -f anatomix_amd/csrc/amx_conv3d_zx.o; make -s -C anatomix_amd/csrc > /dev/null 2>&1
