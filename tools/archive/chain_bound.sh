#!/bin/bash
# conv -> conv chaining at level 0: what would a fused 16 -> 16 pair cost at best?  The z-march kernel with 1, 2 and 3 sweeps per step
# (timing only) = one layer's traffic with 1x / 2x / 3x its matrix work; a real chain needs 2.33x (halo recompute) and its own LDS.
for rep in 0 1 2; do
  AMX_DBG=$((rep << 10)) python tools/layer_profile.py anatomix 4 2>&1 | grep -E "m 3 |m 6 |m62 " | sed "s/^/sweeps=$((rep+1)) /"
done
