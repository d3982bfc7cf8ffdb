#!/bin/bash
# HBM-traffic PMC passes of the headline bench command (GPU box).  FETCH_SIZE and WRITE_SIZE in SEPARATE runs,
# counters only (no trace domains), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
# usage: tools/gpu_pmc.sh <tag>   ->  gpurun_out/<tag>_pmc_traffic.json
TAG=${1:-rXX}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o p -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/${TAG}_pmc_$c.log 2>&1
done
cd $REPO
f=$(ls $OUT/${TAG}_pmc_FETCH_SIZE/*counter_collection.csv | head -1)
w=$(ls $OUT/${TAG}_pmc_WRITE_SIZE/*counter_collection.csv | head -1)
python tools/pmc_summary.py $f $w $OUT/${TAG}_pmc_traffic.json 4
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
