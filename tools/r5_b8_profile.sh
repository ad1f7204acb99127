#!/bin/bash
# kernel statistics of the base config at the reference yaml's 8 images per GPU (imagenet_vitvq_base.yaml:31)
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r5; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_b8 -o st -- python $R/bench.py --batch 8 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r5/bench_b8_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_summary.py stats $(find /tmp/p_b8 -name "*.db" | head -1) $R/gpurun_out/r5/r05_base_b8_kernel_stats.csv | head -45
cd $R
for g in "" "--graphs"; do timeout 300 python bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 $g', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"; done
