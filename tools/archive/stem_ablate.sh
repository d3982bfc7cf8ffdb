#!/bin/bash
# stem A/B + ablations (AMX_DBG 4: no stores; AMX_STEM_WGS: workgroups the launch aims for)
for dbg in 0 4; do for wgs in 256 512 1024; do
  AMX_DBG=$dbg AMX_STEM_WGS=$wgs python tools/stem_time.py 2>&1 | tail -1 | sed "s/^/wgs=$wgs /"
done; done
AMX_STEM_GATHER=1 python tools/stem_time.py 2>&1 | tail -1 | sed "s/^/gather /"
AMX_STEM_GATHER=1 AMX_DBG=4 python tools/stem_time.py 2>&1 | tail -1 | sed "s/^/gather /"
