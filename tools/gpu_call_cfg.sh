#!/bin/bash
export TMPDIR=/tmp
for spec in "imagenet_vitvq_large_full 2" "imagenet_vitvq_large_full 16" "imagenet_vitvq_large 32" "imagenet_rqvae_base 128" "imagenet_vitvq_small 128"; do
  set -- $spec
  timeout 400 python bench.py --config $1 --batch $2 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', 'B=$2', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss', d.get('final_loss'))
except Exception as e: print('$1 B=$2 FAILED', e)"
done
