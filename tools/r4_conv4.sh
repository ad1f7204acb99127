#!/bin/bash
timeout 600 python -m pytest tests/test_conv_nhwc_gpu.py tests/test_disc_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
timeout 200 python tools/conv_bench.py 16 auto 2>/dev/null | head -3
