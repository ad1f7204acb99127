#!/bin/bash
# round-2 evidence in one gpurun call:  gpurun --timeout 1500 -- 'bash tools/profile_round2.sh'
# (1) rocprofv3 --kernel-trace --stats of the bench command ; (2) separate --pmc passes (FETCH_SIZE / WRITE_SIZE) for the HBM traffic of the GEMM kernels ;
# (3) the bench line itself ; (4) the full-loss (LPIPS + GAN) and adversarial configurations + the adversarial step's kernel stats ; (5) per-layer conv timings
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o st -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_summary.py stats $(find /tmp/p_stats -name "*.db" | head -1) $R/gpurun_out/prof/r02_bench_kernel_stats.csv | head -16
timeout 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py pmc $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) $R/gpurun_out/prof/pmc_gemm.json | grep -A3 "true, true, 6"
cd $R
cp gpurun_out/prof/pmc_gemm.json profiles/pmc_gemm.json 2>/dev/null
timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | tee gpurun_out/prof/r02_bench_n1.json | cut -c1-300
timeout 300 python bench.py --config imagenet_vitvq_base_full --batch 16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tee gpurun_out/prof/r02_full_loss_bench.json | cut -c1-200
timeout 300 python bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tee gpurun_out/prof/r02_adv_step_bench.json | cut -c1-200
timeout 300 python bench.py --config imagenet_vitvq_base_adv --batch 64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tee gpurun_out/prof/r02_adv_step_bench_b64.json | cut -c1-200
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_adv -o st -- python $R/bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/p_adv -name "*.db" | head -1) $R/gpurun_out/prof/r02_adv_step_kernel_stats.csv | head -12
cd $R
timeout 200 python tools/conv_bench.py 16 2>/dev/null | tee gpurun_out/prof/r02_conv_layers.txt | tail -3
