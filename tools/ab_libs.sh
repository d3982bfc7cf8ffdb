#!/bin/bash
# A/B: new lib vs lib_old.bin on the TX=32 shapes and the headline
cd $GRAFT_REPO_ROOT
L=anatomix_amd/csrc/libanatomix_amd.so
cp $L /tmp/new.so
for which in new old new old; do
  if [ $which = old ]; then cp anatomix_amd/csrc/lib_old.bin $L; else cp /tmp/new.so $L; fi
  echo "== $which"
  for shape in "32 64 32 64 4" "32 0 32 128 2" "64 0 32 128 2"; do timeout 120 python tools/one_layer.py $shape 2>/dev/null | tail -1; done
  python bench.py --no-secondary --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'])"
  python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 --precision strict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('strict', d['value'])"
done
cp /tmp/new.so $L
timeout 600 python -m pytest tests/test_conv_kernel_gpu.py tests/test_unet_gpu.py tests/test_strict_precision_gpu.py tests/test_unet_dev_gpu.py -x -q -m gpu 2>&1 | tail -2
