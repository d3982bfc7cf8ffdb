#!/bin/bash
for v in "" "AMX_STEM_MFMA=1"; do for wgs in 512 768 1024 2048; do
  env $v AMX_STEM_WGS=$wgs python tools/layer_profile.py anatomix 4 2>&1 | grep -E "m 0 " | sed "s/^/$v wgs=$wgs /"
done; done
