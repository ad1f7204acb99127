#!/bin/bash
# round 5, call 3: the parity_mode block with the x3 producers fused into the GEMM epilogues vs the round-4 two-call form (same box)
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
for f in 1 0; do
ENH_X3_FUSED_SPLIT=$f timeout 900 python bench.py --steps 10 --warmup 3 2>gpurun_out/r5/bench3_$f.err | tee gpurun_out/r5/bench3_fused$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fused=$f bench', d['value'], 'img/s', d['ms_per_step'], 'ms/step', 'vq_match_rate', d.get('vq_match_rate'), 'cpu', d.get('cpu_baseline',{}).get('value'))
pm=d.get('parity_mode',{})
print(json.dumps({k:v for k,v in pm.items() if k!='note' and k!='x3_whole_forward_kernels'}, indent=1))
for r in pm.get('x3_whole_forward_kernels',{}).get('top',[]): print(r)
"
tail -3 gpurun_out/r5/bench3_$f.err | grep -v amdgpu.ids
done
