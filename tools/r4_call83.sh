#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
F="amdgpu.ids\|Warning\|socket.cpp\|version\|Hostname\|Librccl"
{ echo "== ops"; python -m pytest tests/test_ops_gpu.py tests/test_b16_storage_gpu.py -q -p no:cacheprovider 2>&1 | tail -2
  echo "== dp"; python -m pytest tests/test_dp_gpu.py -q -p no:cacheprovider 2>&1 | tail -2
  echo "== dp split"; UZ_CONV_MATH=split python -m pytest tests/test_dp_gpu.py tests/test_unet_probunet_gpu.py -q -p no:cacheprovider 2>&1 | tail -2
  echo "== diag"; UZ_DIAG_STEPS=6 python tools/diag_dp_race.py 2>&1 | grep -v "$F" | cut -c1-200 | head -14
} > gpurun_out/r4_call83.txt 2>&1
