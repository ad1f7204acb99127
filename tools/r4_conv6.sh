#!/bin/bash
timeout 600 python -m pytest tests/test_disc_model_gpu.py tests/test_conv_nhwc_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -s 2>&1 | grep -E "passed|failed|discriminator vs|emulation|Error|assert" | tail -8
for spec in "imagenet_vitvq_base_adv 16" "imagenet_vitvq_base_full 16"; do
  set -- $spec
  for g in "" "--graphs"; do
  timeout 300 python bench.py --config $1 --batch $2 --steps 16 --warmup 17 --no-cpu-baseline --no-parity-mode $g 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', $2, '$g', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
