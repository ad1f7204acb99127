#!/bin/bash
R=$PWD
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_nhwc_gpu.py tests/test_disc_model_gpu.py tests/test_lpips_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3
timeout 300 python tools/conv_bench.py 16 2>&1 | tail -14
timeout 300 python bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:(round(v['total_ms'],1),v['tflops']) for k,v in d['kernels'].items() if 'conv' in k})"
