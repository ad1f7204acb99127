#!/bin/bash
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
timeout 600 python -m pytest tests/test_conv_nhwc_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3
for lib in libenh_hip.so libenh_hip_d1.so; do
  echo "== $lib"
  ENH_HIP_LIB=$R/enhancing-transformers_amd/lib/$lib timeout 300 python bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:(round(v['total_ms'],1),v['tflops']) for k,v in d['kernels'].items() if 'conv' in k})"
done
for b in 32 64; do
  timeout 300 python bench.py --config imagenet_vitvq_base_adv --batch $b --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch', d['config']['per_gpu_batch'], d['value'], d['ms_per_step'], {k:(round(v['total_ms'],1),v['tflops']) for k,v in d['kernels'].items() if 'conv' in k})"
done
