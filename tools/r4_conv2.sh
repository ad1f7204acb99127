#!/bin/bash
# round-4 convolution session 2: end-to-end A/B of the convolution families on the adversarial / full-loss steps + kernel shares
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/prof
timeout 600 python -m pytest tests/test_conv_nhwc_gpu.py tests/test_lpips_gpu.py tests/test_disc_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
for fam in ${FAMS:-t128 auto}; do
  for spec in "imagenet_vitvq_base_adv 16" "imagenet_vitvq_base_adv 64" "imagenet_vitvq_base_full 16"; do
    set -- $spec
    ENH_CONV_KERNEL=$fam timeout 300 python bench.py --config $1 --batch $2 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 > gpurun_out/prof/r04_${1}_b$2_$fam.json
    python -c "
import json,sys
d=json.loads(open('gpurun_out/prof/r04_${1}_b$2_$fam.json').read().strip().splitlines()[-1]); print('$fam', '$1', $2, d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_adv -o st -- python $R/bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/p_adv -name "*.db" | head -1) $R/gpurun_out/prof/r04_adv_step_kernel_stats.csv | head -40
