#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
python -m pytest tests/test_split_storage_gpu.py -q -p no:cacheprovider -k "table" 2>&1 | tail -15 > gpurun_out/r4_call87.txt
