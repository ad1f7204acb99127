#!/bin/bash
timeout 300 python -m pytest tests/test_conv_nhwc_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "blur" 2>&1 | tail -3
timeout 200 python tools/conv_bench.py 16 auto ${VARS:-1,2,4,5,7} 2>/dev/null | grep blur | tee gpurun_out/blur_lab.txt
