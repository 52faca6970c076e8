#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
UZ_OP_PROFILE_BURST=4 UZ_OP_PROFILE_STREAMING=1 python tools/op_profile.py 2>&1 | grep "BN_RELU_FWD.*128, 128, 1, 1\|BN_RELU_FWD.*64, 64, 1, 1\|^total"
echo "== tests"; python -m pytest tests/test_ops_gpu.py tests/test_split_storage_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -2
echo "== bench"; python bench.py 2>/dev/null | tail -1
} > gpurun_out/r4_call75.txt 2>&1
