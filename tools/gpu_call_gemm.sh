#!/bin/bash
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_scale_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "gemm or scale or contraction or epilogue" 2>&1 | tail -4
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tee gpurun_out/prof/ae_bench_try.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(k, round(v['total_ms']/10,2), v['tflops']) for k,v in d['kernels'].items()]"
