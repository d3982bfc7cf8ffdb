#!/bin/bash
# Runs on the GPU box (via gpurun): the full default bench line (headline + secondary workloads), rocprofv3 kernel-trace
# summaries of the headline, strict, dev, ViT and step commands, and the HBM-traffic PMC passes of the headline.
# usage: tools/gpu_round2.sh <tag>      outputs under gpurun_out/<tag>_*
set -u
TAG=${1:-r02}
REPO=$PWD
OUT=$PWD/gpurun_out
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 400 $OUT/${TAG}_bench.json; echo
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_$name -o p -- python $REPO/bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/${TAG}_prof_$name.log 2>&1
  local f=$(ls $OUT/${TAG}_prof_$name/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && head -40 $f > $OUT/${TAG}_bench_${name}_kernel_stats.csv
  rm -rf $OUT/${TAG}_prof_$name
}
prof headline --steps 20
prof strict --precision strict --steps 10 --warmup 3
prof dev --variant anatomix-dev --batch 4 --steps 10 --warmup 3
prof vit --variant anatomix-dev-vit --batch 4 --steps 8 --warmup 3
prof step --workload step --no-graph --steps 5 --warmup 2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o p -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/${TAG}_pmc_$c.log 2>&1
done
cd $REPO
f=$(ls $OUT/${TAG}_pmc_FETCH_SIZE/*counter_collection.csv | head -1)
w=$(ls $OUT/${TAG}_pmc_WRITE_SIZE/*counter_collection.csv | head -1)
python tools/pmc_summary.py $f $w $OUT/${TAG}_pmc_traffic.json 4
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
mv $OUT/${TAG}_bench_headline_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
head -8 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-150
