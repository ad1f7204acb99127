#!/bin/bash
# round-6 evidence in one gpurun call:  gpurun --timeout 1800 -- 'bash tools/profile_round6.sh'
# (0) the whole GPU suite ; (1) rocprofv3 --kernel-trace --stats of the bench command ; (2) separate --pmc passes (FETCH_SIZE / WRITE_SIZE, counters only) over one
# step of the SAME command -> profiles/pmc_step.json keyed by config + batch (bench.py attaches `traffic` only when they match) ; (3) the bench line itself
# (with the parity_mode block and the CPU baseline).
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
if [ "$1" != "nosuite" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/r06_gpu_suite.log | tail -6
fi
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o st -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_summary.py stats $(find /tmp/p_stats -name "*.db" | head -1) $R/gpurun_out/prof/r06_bench_kernel_stats.csv | head -24
timeout 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py pmc $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) $R/gpurun_out/prof/pmc_step.json imagenet_vitvq_base 128
cd $R
cp gpurun_out/prof/pmc_step.json profiles/pmc_step.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/prof/r06_bench.err | tee gpurun_out/prof/r06_bench_n1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms/step; roofline', json.dumps(d['roofline']))
print('parity_mode', json.dumps(d.get('parity_mode'), indent=1))
print('vq_match_rate', d.get('vq_match_rate'), d.get('vq_match_rate_source'), 'cpu', d.get('cpu_baseline',{}).get('value'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['share_of_step'])[:14]: print(f\"{v['share_of_step']:.3f} {v['achieved']:>8} {v['unit']} frac {v['frac']} traffic {v['traffic']}  {k}\")
"
tail -3 gpurun_out/prof/r06_bench.err | grep -v amdgpu.ids
