#!/bin/bash
# round-4 session 2: full GPU suite on the new build, attention family A/B (eight-wave antiphase forward), contention re-measure
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tee gpurun_out/s2_pytest.log | tail -15
PRE=1 ROUNDS=3 timeout 200 python tools/attn_lab.py 1,1,2 4,1,2 2>&1 | tee gpurun_out/s2_attn_pre.txt
PRE=0 ROUNDS=2 timeout 200 python tools/attn_lab.py 1,1,2 4,1,2 2>&1 | tee gpurun_out/s2_attn_plain.txt
timeout 300 python tools/comm_contention.py --ks 0,8,16 2>&1 | tail -30
