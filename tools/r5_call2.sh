#!/bin/bash
# round 5, call 2: new tests (fused x3 split epilogues, ADVICE regressions, bench self-launch), tile-order lab with the per-shape default, bench with parity_mode
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_x3_gpu.py tests/test_gumbel_gpu.py tests/test_tools_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -s 2>&1 | grep -v amdgpu.ids | tail -40 | tee gpurun_out/r5/call2_tests.log
LAB_EXP=order timeout 300 python tools/gemm_ld_lab.py 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/gemm_order_lab2.txt
for f in 1 0; do
ENH_X3_FUSED_SPLIT=$f timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/r5/bench2_$f.err | tee gpurun_out/r5/bench2_fused$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fused=$f bench', d['value'], 'img/s', d['ms_per_step'], 'ms/step')
pm=d.get('parity_mode',{})
print(json.dumps({k:v for k,v in pm.items() if k!='note' and k!='x3_whole_forward_kernels'}, indent=1))
for r in pm.get('x3_whole_forward_kernels',{}).get('top',[]): print(r)
"
done
