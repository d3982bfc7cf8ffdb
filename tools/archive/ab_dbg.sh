for d in 0 256 0 256; do
  echo "== AMX_DBG=$d"
  for shape in "32 64 32 64 4" "64 0 32 128 2"; do AMX_DBG=$d timeout 120 python tools/one_layer.py $shape 2>/dev/null | tail -1; done
  AMX_DBG=$d python bench.py --no-secondary --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'])"
  AMX_DBG=$d python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 --precision strict 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('strict', d['value'])"
  AMX_DBG=$d python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 --variant anatomix-dev --batch 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dev', d['value'])"
done
timeout 600 python -m pytest tests/test_conv_kernel_gpu.py tests/test_unet_gpu.py tests/test_strict_precision_gpu.py tests/test_unet_dev_gpu.py -x -q -m gpu 2>&1 | tail -2
