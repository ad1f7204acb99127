#!/bin/bash
# validation + measurement of the implicit-GEMM discriminator path (one gpurun call)
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
timeout 900 python -m pytest tests/test_conv_nhwc_gpu.py tests/test_disc_model_gpu.py tests/test_lpips_gpu.py -m gpu -q --no-header -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -60 > $R/gpurun_out/disc_tests.log
tail -25 $R/gpurun_out/disc_tests.log
timeout 300 python bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/prof/adv_bench.json | cut -c1-300
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_adv -o st -- python $R/bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/p_adv -name "*.db" | head -1) $R/gpurun_out/prof/adv_kernel_stats.csv | head -24
