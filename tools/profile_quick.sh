#!/bin/bash
# quick regression + bench + kernel stats (one gpurun call)
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_parity_base_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -3
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o st -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_summary.py stats $(find /tmp/p_stats -name "*.db" | head -1) $R/gpurun_out/prof/r02_bench_kernel_stats.csv | head -14
cd $R
timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | tee gpurun_out/prof/r02_bench_n1.json | cut -c1-200
