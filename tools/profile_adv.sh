#!/bin/bash
# kernel breakdown of the adversarial and full-loss steps (one gpurun call)
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
cd /tmp
for cfg in imagenet_vitvq_base_adv imagenet_vitvq_base_full; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$cfg -o st -- python $R/bench.py --config $cfg --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/${cfg}_under_rocprof.json 2>/dev/null
  python $R/tools/rocpd_summary.py stats $(find /tmp/p_$cfg -name "*.db" | head -1) $R/gpurun_out/prof/r02_${cfg}_kernel_stats.csv | head -30
done
cd $R
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-400
