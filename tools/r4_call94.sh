#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{ python -m pytest tests/test_phiseg_gpu.py tests/test_unet_probunet_gpu.py -q -p no:cacheprovider -k "two_replayed or deferred_tables or graph_replay or probunet" 2>&1 | tail -12
} > gpurun_out/r4_call94.txt 2>&1
