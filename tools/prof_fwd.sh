#!/bin/bash
# rocprofv3 kernel stats of the plain forward loop (bench.py headline, no secondaries): usage tools/prof_fwd.sh <tag> [bench args]
TAG=${1:-rXX}; shift
OUT=$PWD/gpurun_out; REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python $REPO/bench.py --steps 20 --no-cpu-baseline --no-secondary --no-parity "$@" > $OUT/${TAG}_prof.log 2>&1
cd $REPO
f=$(ls $OUT/${TAG}_prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv && python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:28]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
tail -c 400 $OUT/${TAG}_prof.log
