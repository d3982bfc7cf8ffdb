#!/bin/bash
# A/B of two library builds on ONE box: anatomix_amd/csrc/libanatomix_amd.so (new) against anatomix_amd/csrc/lib_old.bin (built from
# another revision and copied there by hand; not tracked).  usage (on the GPU box): bash tools/ab_libs.sh
cd $GRAFT_REPO_ROOT
L=anatomix_amd/csrc/libanatomix_amd.so
cp $L /tmp/new.so
for which in new old new old; do
  if [ $which = old ]; then cp anatomix_amd/csrc/lib_old.bin $L; else cp /tmp/new.so $L; fi
  python bench.py --no-secondary --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']; print('$which headline', d['value'], [(k[9:22]+k[-12:],v['us_per_step']) for k,v in list(pk.items())[:9]])"
done
cp /tmp/new.so $L
