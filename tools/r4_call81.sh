#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{ python tools/diag_dp_race.py 2>&1 | grep -v "amdgpu.ids\|Warning\|socket.cpp"; echo "== again"; python tools/diag_dp_race.py 2>&1 | grep -v "amdgpu.ids\|Warning\|socket.cpp"; } > gpurun_out/r4_call81.txt 2>&1
