cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "gemm" 2>&1 | tail -8
timeout 300 python -m pytest tests/test_scale_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k gemm 2>&1 | grep -E "rel |passed|failed|Error" | cut -c1-200
MB_BATCH=128 timeout 200 python tools/microbench.py 2>/dev/null | sed -n 1,14p
