#!/bin/bash
# Runs on the GPU box (via gpurun): the full default bench line, rocprofv3 kernel-trace summaries of the headline / anatomix-dev
# (f16x2mx) / strict / step commands, per-layer tables, HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) of the
# headline AND of the anatomix-dev forward, SQ counters of both, the step timeline.
# usage: tools/gpu_round6.sh <tag>      outputs under gpurun_out/<tag>_*
set -u
TAG=${1:-r06}
REPO=$PWD
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1000 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cp $OUT/bench_full.json $OUT/${TAG}_bench_full.json 2>/dev/null
wc -c $OUT/${TAG}_bench.json; tail -c 700 $OUT/${TAG}_bench.json; echo
(timeout 300 python tools/layer_profile.py anatomix 4) > $OUT/${TAG}_layers_6m.txt 2>/dev/null
(timeout 300 python tools/layer_profile.py anatomix-dev 4) > $OUT/${TAG}_layers_dev.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_$name -o p -- python $REPO/bench.py --no-cpu-baseline --no-secondary --no-parity --sustain 0 "$@" > $OUT/${TAG}_prof_$name.log 2>&1
  local f=$(ls $OUT/${TAG}_prof_$name/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && head -40 $f > $OUT/${TAG}_bench_${name}_kernel_stats.csv
  rm -rf $OUT/${TAG}_prof_$name
}
prof headline --steps 20
prof dev --variant anatomix-dev --batch 4 --steps 8 --warmup 2
prof strict --precision strict --steps 10 --warmup 3
prof step --workload step --no-graph --steps 5 --warmup 2
pmc() {  # name, batch, bench args...
  local name=$1 batch=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc_${name}_$c -o p -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-parity --sustain 0 "$@" > $OUT/${TAG}_pmc_${name}_$c.log 2>&1
  done
  local f=$(ls $OUT/${TAG}_pmc_${name}_FETCH_SIZE/*counter_collection.csv | head -1)
  local w=$(ls $OUT/${TAG}_pmc_${name}_WRITE_SIZE/*counter_collection.csv | head -1)
  (cd $REPO && python tools/pmc_summary.py $f $w $OUT/${TAG}_${name}pmc_traffic.json $batch)
  rm -rf $OUT/${TAG}_pmc_${name}_FETCH_SIZE $OUT/${TAG}_pmc_${name}_WRITE_SIZE
}
pmc "" 4
pmc dev_ 4 --variant anatomix-dev --batch 4
cd $REPO
mv $OUT/${TAG}_bench_headline_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
head -12 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-150
# SQ-side counters (separate passes, counters only)
timeout 500 tools/gpu_pmc_sq.sh ${TAG} --sustain 0 > $OUT/${TAG}_pmc_sq.txt 2>&1
tail -14 $OUT/${TAG}_pmc_sq.txt
timeout 500 tools/gpu_pmc_sq.sh ${TAG}_dev --sustain 0 --variant anatomix-dev > $OUT/${TAG}_dev_pmc_sq.txt 2>&1
tail -12 $OUT/${TAG}_dev_pmc_sq.txt
timeout 400 bash tools/step_timeline.sh ${TAG} > /dev/null 2>&1
head -12 $OUT/${TAG}_step_timeline.txt
