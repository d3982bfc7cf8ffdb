#!/bin/bash
# Runs on the GPU box (via gpurun): headline bench + kernel-trace profile + extra-config benches.
# usage: tools/gpu_round.sh <tag>      outputs under gpurun_out/<tag>_*
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
python bench.py --variant anatomix-dev --batch 2 --steps 20 --cpu-forwards 2 > $OUT/${TAG}_bench_dev.json 2> $OUT/${TAG}_bench_dev.err
python bench.py --sw-volume 256 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_sw256.json 2> $OUT/${TAG}_bench_sw256.err
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python $REPO/bench.py --steps 20 --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1
cd $REPO
f=$(ls $OUT/${TAG}_prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_bench_kernel_stats.csv && head -12 $f
python - <<PY
import json
for n in ("bench","bench_dev","bench_sw256"):
    try:
        d=json.load(open("$OUT/${TAG}_%s.json"%n)); print(n, d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel"][:60], d["roofline"]["frac"], d.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(n, "FAILED", e)
PY
