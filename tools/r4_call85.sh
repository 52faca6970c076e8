#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
F="amdgpu.ids\|Warning\|socket.cpp\|version\|Hostname\|Librccl"
{ echo "== dp tests"; python -m pytest tests/test_dp_gpu.py -q -p no:cacheprovider 2>&1 | tail -3
  for t in 1 0 1 0; do echo "== world-1 check, UZ_DP_TABLES=$t"; UZ_DP_TABLES=$t MASTER_PORT=2957$t python tools/nccl_world1_check.py 2>&1 | grep "no dp\|bit-identical\|buckets"; done
} > gpurun_out/r4_call85.txt 2>&1
