#!/bin/bash
# the configs beside the headline, one line each (profiles/r04_other_configs.md)
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "head_scaled" 2>&1 | tail -2
( for spec in "imagenet_vitvq_small 128" "imagenet_rqvae_base 128" "imagenet_vitvq_large 32" "imagenet_vitvq_large_full 16" "imagenet_vitvq_base_adv 64"; do
  set -- $spec
  timeout 400 python bench.py --config $1 --batch $2 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$1', 'B=$2', d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant kernel', r.get('kernel'), r.get('achieved'), r.get('unit'))
except Exception as e: print('$1 B=$2 FAILED', e)"
done ) | tee gpurun_out/r04_other_configs_raw.txt
