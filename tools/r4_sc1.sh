#!/bin/bash
export TMPDIR=/tmp
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], 'img/s', d['ms_per_step'], 'ms')
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['share_of_step'])[:10]: print(f\"{v['share_of_step']:.3f} {v['achieved']:>8} {v['unit']}  {k}\")"
timeout 200 python tools/conv_bench.py 16 auto 2>/dev/null | grep -v blur
timeout 300 python bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 8 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('adv16', d['value'], 'img/s', d['ms_per_step'], 'ms')"
