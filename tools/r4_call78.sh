#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
for v in base nt base nt; do
  if [ $v = nt ]; then export UZ_LIB=$PWD/tools/tmp/libuz_hip_nt.so; else unset UZ_LIB; fi
  echo "== $v"; UZ_OP_PROFILE_BURST=4 UZ_OP_PROFILE_STREAMING=1 python tools/op_profile.py 2>&1 | grep "BN_RELU_FWD.*128, 128, 128, 32, 128, 128\|BN_RELU_FWD.*\[64, 64, 64, 32, 64, 64" | head -4
  python bench.py --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | cut -c1-130
done
} > gpurun_out/r4_call78.txt 2>&1
