#!/bin/bash
# builds tools/probe/gemm_lab (needs enhancing-transformers_amd/lib/libenh_hip.so); temps go to /tmp/lab_tmp
set -e
cd "$(dirname "$0")"
mkdir -p /tmp/lab_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps=obj gemm_lab.cpp -o /tmp/lab_tmp/gemm_lab -L ../../enhancing-transformers_amd/lib -lenh_hip '-Wl,-rpath,$ORIGIN/../../enhancing-transformers_amd/lib' 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^" | head -20
cp /tmp/lab_tmp/gemm_lab ./gemm_lab
