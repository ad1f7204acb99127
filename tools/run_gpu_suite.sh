#!/bin/bash
# full GPU regression + per-shape table + the headline bench, one gpurun call:  gpurun --timeout 1500 -- 'bash tools/run_gpu_suite.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -6
MB_BATCH=128 timeout 200 python tools/microbench.py 2>/dev/null | sed -n 1,16p
timeout 400 python bench.py --steps 10 --warmup 3 2>/dev/null | tee gpurun_out/bench_latest.json | cut -c1-400
