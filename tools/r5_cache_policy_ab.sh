#!/bin/bash
# same-box interleaved A/B of cache-policy variants of the persistent GEMMs (lab builds of gemm.o: -DENH_P_NT_STORE=1 / -DENH_A_NT=1, see gemm.hip):
# full bench, default library vs nt result stores vs nt A requests vs both
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
for i in 1 2; do
  for lib in libenh_hip.so libenh_hip_nts.so libenh_hip_ant.so libenh_hip_both.so; do
    [ -f $R/enhancing-transformers_amd/lib/$lib ] || continue
    ENH_HIP_LIB=$R/enhancing-transformers_amd/lib/$lib timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
row=lambda n: next((f\"{v['total_ms']/v['launches']:.4f}\" for kk,v in k.items() if kk.startswith(n)), '-')
print('$lib', d['value'], 'img/s', d['ms_per_step'], 'ms/step | qkv fwd', row('gemm_bf16_w256r_kernel<false, 1'), 'fc1+tanh', row('gemm_bf16_w256r_kernel<false, 2'), 'dgrad bf16', row('gemm_bf16_w256r_kernel<true, 1'), 'bias+res', row('gemm_bf16_w256p_kernel<false, false, 4'), 'dtanh', row('gemm_bf16_w256p_kernel<false, true, 3'), 'wgrad', row('gemm_bf16_w256_kernel<true, true, 6'), 'attn fwd', row('attn_fwd'), 'ln_bwd', row('ln_bwd'))" | tee -a gpurun_out/r5/cache_policy_ab.txt
  done
done
