#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
echo "== tests"; python -m pytest tests/test_ops_gpu.py tests/test_b16_storage_gpu.py tests/test_phiseg3d.py -q -p no:cacheprovider -x -m gpu 2>&1 | tail -3
echo "== bench 3d"; python bench.py --model phiseg3d 2>/dev/null | tail -1
} > gpurun_out/r4_call77.txt 2>&1
