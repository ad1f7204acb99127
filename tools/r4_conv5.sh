#!/bin/bash
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/prof
timeout 200 python tools/conv_bench.py 16 auto 2>/dev/null | tee gpurun_out/conv_layers_auto.txt
for spec in "imagenet_vitvq_base_adv 16" "imagenet_vitvq_base_adv 64" "imagenet_vitvq_base_full 16"; do
  set -- $spec
  timeout 300 python bench.py --config $1 --batch $2 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 > gpurun_out/prof/r04_${1}_b$2_auto.json
  python -c "
import json,sys
d=json.loads(open('gpurun_out/prof/r04_${1}_b$2_auto.json').read().strip().splitlines()[-1]); print('auto', '$1', $2, d['value'], 'img/s', d['ms_per_step'], 'ms')"
done
