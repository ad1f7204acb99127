#!/bin/bash
# First GPU call of round 2: validates the kernels written blind at the end of round 1 and A/B-measures them, in ONE gpurun call.
#   gpurun --timeout 900 -- 'bash tools/round2_first_call.sh 2>&1 | tail -120'
# Every step has its own timeout; nothing here changes defaults.  Output lands in gpurun_out/round2_first/.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/round2_first
mkdir -p "$OUT"
echo "== 1. regression (default kernels) =="
echo "(skipped: green in GPUTEST_r01)"
echo "== 2. experimental 8-phase GEMM: correctness (GEMM suite re-run under ENH_GEMM_KERNEL=8phase), three times as a race screen =="
for i in 1 2 3; do
  ENH_TEST_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "p8 and 8phase" 2>&1 | tail -2
done
ENH_TEST_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "p8 and 9persist" 2>&1 | tail -2
echo "== 2b. bench-scale spot checks (M = 131072): default | 8phase | 9persist =="
timeout 300 python -m pytest tests/test_scale_gpu.py -m gpu -q --no-header -p no:cacheprovider -s 2>&1 | grep -E "rel |attention at|passed|failed|Error" | cut -c1-200
ENH_GEMM_KERNEL=8phase timeout 300 python -m pytest tests/test_scale_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k gemm 2>&1 | grep -E "rel |passed|failed|Error" | cut -c1-200
ENH_GEMM_KERNEL=9persist timeout 300 python -m pytest tests/test_scale_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k gemm 2>&1 | grep -E "rel |passed|failed|Error" | cut -c1-200
echo "== 3. per-shape GEMM table: default | 8phase | 9persist (B = 128) =="
MB_BATCH=128 timeout 120 python tools/microbench.py 2>/dev/null | sed -n 2,14p > "$OUT/gemm_default.txt"
ENH_GEMM_KERNEL=8phase MB_BATCH=128 timeout 120 python tools/microbench.py 2>/dev/null | sed -n 2,14p > "$OUT/gemm_p8.txt"
ENH_GEMM_KERNEL=9persist MB_BATCH=128 timeout 120 python tools/microbench.py 2>/dev/null | sed -n 2,14p > "$OUT/gemm_p8p.txt"
paste -d'|' <(cut -c1-34,68-90 "$OUT/gemm_default.txt") <(cut -c68-90 "$OUT/gemm_p8.txt") <(cut -c68-90 "$OUT/gemm_p8p.txt")
echo "== 4. whole step: default vs 8phase vs 9persist (wherever they apply) =="
timeout 150 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-130
ENH_GEMM_KERNEL=8phase timeout 150 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-130
ENH_GEMM_KERNEL=9persist timeout 150 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-130
echo "== 5. experimental discriminator kernels: correctness, then the adversarial step with and without them =="
ENH_TEST_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_disc_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -k fast_paths 2>&1 | tail -2
timeout 150 python bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-130
ENH_DISC_FAST=1 timeout 150 python bench.py --config imagenet_vitvq_base_adv --batch 16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-130
