#!/bin/bash
# round-3 evidence in one gpurun call:  gpurun --timeout 1500 -- 'bash tools/profile_round3.sh'
# (1) rocprofv3 --kernel-trace --stats of the bench command ; (2) separate --pmc passes (FETCH_SIZE / WRITE_SIZE, counters only) over one step of the SAME
# command -> profiles/pmc_step.json keyed by config + batch (bench.py attaches `traffic` only when they match) ; (3) the bench line itself.
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o st -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_summary.py stats $(find /tmp/p_stats -name "*.db" | head -1) $R/gpurun_out/prof/r03_bench_kernel_stats.csv | head -20
timeout 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py pmc $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) $R/gpurun_out/prof/pmc_step.json imagenet_vitvq_base 128
cd $R
cp gpurun_out/prof/pmc_step.json profiles/pmc_step.json 2>/dev/null
timeout 500 python bench.py --steps 20 --warmup 5 2>/dev/null | tee gpurun_out/prof/r03_bench_n1.json | cut -c1-400
