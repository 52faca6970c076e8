#!/bin/bash
# bilinear backward: float4-per-lane kernel vs the pair kernel; ops tests; bench line with burst family timing
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
echo "== quad"; python tools/bench_resample.py 2>&1 | grep bilinear
echo "== pair"; UZ_BILINEAR_BWD_PAIR=1 python tools/bench_resample.py 2>&1 | grep bilinear
echo "== tests"; python -m pytest tests/test_ops_gpu.py tests/test_split_storage_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -3
echo "== bench"; python bench.py 2>/dev/null | tail -1
} > gpurun_out/r4_call73.txt 2>&1
