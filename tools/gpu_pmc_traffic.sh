#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate counter-only passes) per kernel of a bench command.
# usage: tools/gpu_pmc_traffic.sh <tag> [bench args]   ->  gpurun_out/<tag>_pmc_traffic.json
TAG=${1:-rXX}; shift
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o p -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-parity "$@" > $OUT/${TAG}_pmc_$c.log 2>&1
done
cd $REPO
f=$(ls $OUT/${TAG}_pmc_FETCH_SIZE/*counter_collection.csv | head -1)
w=$(ls $OUT/${TAG}_pmc_WRITE_SIZE/*counter_collection.csv | head -1)
python tools/pmc_summary.py $f $w $OUT/${TAG}_pmc_traffic.json 4
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
