#!/bin/bash
# round-4 session 4: antiphase dK/dV kernel
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -k "attention" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/s4_pytest.log | tail -8
python -c "
import sys; sys.path.insert(0,'enhancing-transformers_amd')
from enhancing import _C
print('wave->SIMD map (workgroup 0 | last workgroup):', _C.wave_simd_map())"
PRE=1 ROUNDS=3 timeout 200 python tools/attn_lab.py 1,1,2 4,1,3 2>&1 | grep family | tee gpurun_out/s4_attn_pre.txt
