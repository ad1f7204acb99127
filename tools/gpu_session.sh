#!/bin/bash
# One parametrised GPU-box script for the recurring measurement sessions (replaces the round-4 one-offs tools/r4_*.sh and the round-5 call scripts).
#   gpurun --timeout 2400 -- 'bash tools/gpu_session.sh <task> [args]'
# tasks
#   suite                     the whole GPU suite                                   -> gpurun_out/gpu_suite.log
#   bench [bench.py args]     the headline line (with parity_mode + cpu_baseline)   -> gpurun_out/bench_latest.json
#   lib-ab LIB...             same-box interleaved A/B of library builds (names under enhancing-transformers_amd/lib/, e.g. libenh_hip.so libenh_hip_nts.so):
#                             headline + per-kernel launch averages, two rounds     -> gpurun_out/lib_ab.txt
#   landing-lab [batch]       leading-dimension + tile-order lab (tools/gemm_ld_lab.py)
#   landing-pmc               TCC / TCP counter passes over tools/gemm_landing_pmc.py
#   parity-ab                 parity_mode block with / without the fused x3 producers (ENH_X3_FUSED_SPLIT)
#   attn-lab FAM...           tools/attn_lab.py (pre-scaled q) for kernel families "fwd,dq,dkv" + the bit-reproducibility probe
#   configs SPEC...           other configs, SPEC = config:batch[:graphs]                -> gpurun_out/other_configs_raw.txt
#   kstats CONFIG BATCH       rocprofv3 --kernel-trace --stats of `bench.py --config CONFIG --batch BATCH` -> gpurun_out/prof/<CONFIG>_b<BATCH>_kernel_stats.csv
#   conv-layers [batch] [fam...]  per-layer convolution table (tools/conv_bench.py)
cd "$(dirname "$0")/.."
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/prof
task=$1; shift
line() {   # one bench line -> short summary; $1 = label
  python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms/step; dominant', r.get('kernel'), r.get('achieved'), r.get('unit'), 'frac', r.get('frac'))
except Exception as e: print('$1 FAILED', e)"
}
case $task in
suite)
  timeout 2000 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/gpu_suite.log | tail -8 ;;
bench)
  timeout 900 python bench.py "$@" 2>gpurun_out/bench_latest.err | tee gpurun_out/bench_latest.json | line bench ;;
lib-ab)
  for i in 1 2; do for lib in "$@"; do
    [ -f $R/enhancing-transformers_amd/lib/$lib ] || { echo "missing $lib"; continue; }
    ENH_HIP_LIB=$R/enhancing-transformers_amd/lib/$lib timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
row=lambda n: next((f\"{v['total_ms']/v['launches']:.4f}\" for kk,v in k.items() if kk.startswith(n)), '-')
print('$lib', d['value'], 'img/s', d['ms_per_step'], 'ms/step | qkv fwd', row('gemm_w256r_kernel<F16, false, 1'), 'fc1+tanh', row('gemm_w256r_kernel<F16, false, 2'), 'dgrad 16-bit', row('gemm_w256r_kernel<F16, true, 1'), 'bias+res', row('gemm_w256p_kernel<F16, false, false, 4'), 'dtanh', row('gemm_w256p_kernel<F16, false, true, 3'), 'wgrad', row('gemm_w256_kernel<F16, true, true, 6'), 'attn fwd', row('attn_fwd'), 'attn bwd', row('attn_bwd'), 'ln_bwd', row('ln_bwd'))" | tee -a gpurun_out/lib_ab.txt
  done; done ;;
landing-lab)
  timeout 600 python tools/gemm_ld_lab.py ${1:-128} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_landing_lab.txt ;;
landing-pmc)
  bash tools/r5_pmc_landing.sh ;;
parity-ab)
  for f in 1 0; do
    ENH_X3_FUSED_SPLIT=$f timeout 900 python bench.py --steps 10 --warmup 3 2>/dev/null | tee gpurun_out/bench_parity_fused$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pm=d.get('parity_mode',{})
print('fused=$f headline', d['value'], json.dumps({k:v for k,v in pm.items() if k not in ('note','x3_whole_forward_kernels')}))
for r in pm.get('x3_whole_forward_kernels',{}).get('top',[]): print('   ', r)"
  done ;;
attn-lab)
  PRE=1 ROUNDS=${ROUNDS:-5} python tools/attn_lab.py "$@" 2>&1 | grep family
  python tools/attn_det_probe.py 2>&1 | grep -v amdgpu.ids ;;
configs)
  ( for spec in "$@"; do
      IFS=: read cfg b g <<< "$spec"
      timeout 400 python bench.py --config $cfg --batch $b --steps ${STEPS:-8} --warmup ${WARMUP:-3} --no-cpu-baseline --no-parity-mode ${g:+--graphs} 2>/dev/null | line "$cfg B=$b ${g:+graphs}"
    done ) | tee -a gpurun_out/other_configs_raw.txt ;;
kstats)
  cd /tmp; rm -rf /tmp/p_ks
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_ks -o st -- python $R/bench.py --config $1 --batch $2 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
  python $R/tools/rocpd_summary.py stats $(find /tmp/p_ks -name "*.db" | head -1) $R/gpurun_out/prof/$1_b$2_kernel_stats.csv | head -45 ;;
conv-layers)
  b=${1:-16}; shift
  for fam in ${@:-t128 auto}; do timeout 200 python tools/conv_bench.py $b $fam 2>/dev/null | tee gpurun_out/conv_layers_$fam.txt; done ;;
*) echo "unknown task $task"; exit 2 ;;
esac
