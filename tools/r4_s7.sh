#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tools_gpu.py tests/test_disc_model_gpu.py tests/test_x3_gpu.py tests/test_gumbel_gpu.py -m gpu -q --no-header -p no:cacheprovider -s 2>&1 | grep -E "passed|failed|full x3|train step|tiny vs|Error|assert" | tail -12
for spec in "imagenet_vitvq_large_full 2" "imagenet_vitvq_base_full 16"; do
  set -- $spec
  timeout 300 python bench.py --config $1 --batch $2 --steps 16 --warmup 17 --no-cpu-baseline --graphs 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', 'B=$2', '--graphs', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss', d.get('final_loss'))
except Exception as e: print('$1 B=$2 FAILED', e)"
done
