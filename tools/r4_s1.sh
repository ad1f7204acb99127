#!/bin/bash
# round-4 session 1: x3 path + dynamic schedule + RCCL smoke tests, contention experiment, bench with the parity_mode block
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_x3_gpu.py tests/test_nccl_gpu.py "tests/test_ops_gpu.py::test_persistent_gemm_is_bitwise_the_one_tile_kernel" "tests/test_ops_gpu.py::test_dynamic_tile_schedule_under_cu_contention" "tests/test_ops_gpu.py::test_persistent_gemm_with_row_strides" -m gpu -q --no-header -p no:cacheprovider -s 2>&1 | tee gpurun_out/s1_pytest.log | grep -v "^$" | tail -40
timeout 300 python tools/comm_contention.py 2>&1 | tail -40
timeout 400 python bench.py --steps 8 --warmup 2 2>gpurun_out/s1_bench.err | tee gpurun_out/s1_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], 'img/s', d['ms_per_step'], 'ms/step; roofline', d['roofline']['kernel'], d['roofline']['frac'])
print('parity_mode', json.dumps(d.get('parity_mode'), indent=1))
print('vq_match_rate', d.get('vq_match_rate'), d.get('vq_match_rate_source'))
"
tail -5 gpurun_out/s1_bench.err
