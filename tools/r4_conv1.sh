#!/bin/bash
# round-4 convolution session: parity of the 256-row kernels, then the per-layer table under each family
timeout 600 python -m pytest tests/test_conv_nhwc_gpu.py tests/test_lpips_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -15
mkdir -p gpurun_out
for fam in ${FAMS:-t128 auto}; do timeout 200 python tools/conv_bench.py 16 $fam 2>/dev/null | grep -v blur | tee gpurun_out/conv_layers_$fam.txt; done
