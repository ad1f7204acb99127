#!/bin/bash
# where the time of the shipped small-batch configs goes: eager kernel shares (rocprofv3) + eager / graph throughput
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/prof
for spec in "imagenet_vitvq_large_full 2" "imagenet_vitvq_base_full 16"; do
  set -- $spec
  for g in "" "--graphs"; do
    timeout 300 python bench.py --config $1 --batch $2 --steps 16 --warmup 17 --no-cpu-baseline --no-parity-mode $g 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', 'B=$2', '$g', d['value'], 'img/s', d['ms_per_step'], 'ms')
except Exception as e: print('$1 B=$2 FAILED', e)"
  done
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_lf -o st -- python $R/bench.py --config imagenet_vitvq_large_full --batch 2 --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/p_lf -name "*.db" | head -1) $R/gpurun_out/prof/r04_large_full_b2_kernel_stats.csv | head -45
