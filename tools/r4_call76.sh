#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
UZ_OP_PROFILE_JSON=gpurun_out/r4_op_times_burst.json UZ_OP_PROFILE_BURST=4 python tools/op_profile.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tail -12 > gpurun_out/r4_call76.txt
