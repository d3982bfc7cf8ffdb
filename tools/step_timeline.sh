#!/bin/bash
# kernel timeline of the graph-replayed contrastive step -> gpurun_out/<tag>_step_timeline.txt
TAG=${1:-r03}; REPO=$(cd $(dirname $0)/.. && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_step_trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_step_trace -o t -- python $REPO/bench.py --workload step --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-parity --sustain 0 > $OUT/${TAG}_step_trace.log 2>&1
f=$(find $OUT/${TAG}_step_trace -name '*kernel_trace.csv' | head -1)
python $REPO/tools/step_timeline.py $f $OUT/${TAG}_step_listing.txt > $OUT/${TAG}_step_timeline.txt 2>&1
tail -5 $OUT/${TAG}_step_trace.log
cat $OUT/${TAG}_step_timeline.txt
cp $f $OUT/${TAG}_step_kernel_trace.csv; rm -rf $OUT/${TAG}_step_trace
