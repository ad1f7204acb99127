#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_base_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "reproducible or layernorm or colsum or golden or train" 2>&1 | tail -2
