#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <python args...>   -- rocprofv3 kernel-trace stats of one python command (GPU box)
TAG=$1; shift
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python "$@" > $OUT/${TAG}_prof.log 2>&1
cd $REPO
f=$(ls $OUT/${TAG}_prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv && head -${LINES_OUT:-14} $f | cut -c1-220
