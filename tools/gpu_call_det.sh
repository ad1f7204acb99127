#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
for i in 1 2; do timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
