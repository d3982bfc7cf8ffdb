#!/bin/bash
# rocprofv3 kernel-trace summary of the headline forward (GPU box).  usage: tools/prof_headline.sh <tag> [bench args...]
TAG=${1:-t}; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python $REPO/bench.py --no-cpu-baseline --no-secondary --no-parity --sustain 0 --steps 20 "$@" > $OUT/${TAG}_prof.log 2>&1
f=$(ls $OUT/${TAG}_prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -40 $f > $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_prof
tail -1 $OUT/${TAG}_prof.log | cut -c1-300
cut -d, -f1-4 $OUT/${TAG}_kernel_stats.csv | head -24 | cut -c1-190
