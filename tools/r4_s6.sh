#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -k "vq" "tests/test_parity_base_gpu.py::test_training_step_gradients_are_bit_reproducible" tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -5
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms/step')
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['share_of_step'])[:12]: print(f\"{v['share_of_step']:.3f} {v['achieved']:>8} {v['unit']} {v['total_ms']/v['launches']:.4f} ms  {k}\")
"
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o st -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1; python $OLDPWD/tools/rocpd_summary.py stats $(find /tmp/p2 -name "*.db" | head -1) /tmp/x.csv | grep -i "vq_\|attn_fwd"
