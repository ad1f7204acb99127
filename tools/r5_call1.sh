#!/bin/bash
# round 5, call 1: (a) leading-dimension + tile-order lab, (b) same-box interleaved A/B of the atomic-optimizer flag
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
timeout 600 python tools/gemm_ld_lab.py 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/gemm_landing_lab.txt
for i in 1 2 3; do
  for lib in libenh_hip.so libenh_hip_noflag.so; do
    ENH_HIP_LIB=$R/enhancing-transformers_amd/lib/$lib timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], 'img/s', d['ms_per_step'], 'ms/step')" | tee -a gpurun_out/r5/flag_ab.txt
  done
done
