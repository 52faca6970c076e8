#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
F="amdgpu.ids\|Warning\|socket.cpp\|version\|Hostname\|Librccl"
{ echo "== bits quad"; python tools/diag_bilinear_bits.py 2>&1 | grep -v "$F"; echo "== bits pair"; UZ_BILINEAR_BWD_PAIR=1 python tools/diag_bilinear_bits.py 2>&1 | grep -v "$F"
  echo "== graphs"; UZ_DIAG_STEPS=4 python tools/diag_dp_race.py 2>&1 | grep -v "$F" | cut -c1-200 | head -12
  echo "== eager"; UZ_DIAG_EAGER=1 UZ_DIAG_STEPS=4 python tools/diag_dp_race.py 2>&1 | grep -v "$F" | cut -c1-200 | head -12
} > gpurun_out/r4_call82.txt 2>&1
