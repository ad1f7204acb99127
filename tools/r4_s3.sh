#!/bin/bash
# round-4 session 3: antiphase attention v3.1 (fragments fetched in the vector segment), graph replay of the two-optimizer step
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -k "attention" "tests/test_disc_model_gpu.py::test_graph_replay_of_the_two_optimizer_step_equals_the_eager_sequence" -m gpu -q --no-header -p no:cacheprovider 2>&1 | tee gpurun_out/s3_pytest.log | tail -25
PRE=1 ROUNDS=3 timeout 200 python tools/attn_lab.py 1,1,2 4,1,2 2>&1 | grep family | tee gpurun_out/s3_attn_pre.txt
for spec in "imagenet_vitvq_large_full 2" "imagenet_vitvq_base_full 16"; do
  set -- $spec
  for g in "" "--graphs"; do
    timeout 300 python bench.py --config $1 --batch $2 --steps 16 --warmup 17 --no-cpu-baseline $g 2>gpurun_out/s3_bench_err.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', 'B=$2', '$g', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss', d.get('final_loss'))
except Exception as e: print('$1 B=$2 $g FAILED', e)"
    tail -3 gpurun_out/s3_bench_err.txt | grep -v amdgpu.ids
  done
done
