#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
UZ_OP_PROFILE_BURST=4 UZ_OP_PROFILE_STREAMING=1 python tools/op_profile.py 2>&1 | grep -v "amdgpu.ids\|Warning" | head -90 > gpurun_out/r4_call74.txt
