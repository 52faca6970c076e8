#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export UZ_DIAG_STEPS=6 UZ_DIAG_NETS=o,o
run() { python tools/diag_dp_race.py 2>/dev/null | grep "^step.*params" | cut -c1-100 | grep -c "differ: 0"; }
{ for cfg in "UZ_LANES=3" "UZ_DP_STREAM_PRIORITY=0" "UZ_DIAG_BATCH=16" "UZ_DIAG_BATCH=8" "UZ_DP_TABLES=0 UZ_DIAG_BATCH=16" "UZ_DP_TABLES=0 UZ_LANES=3" "UZ_CONV_MATH=split" "UZ_CONV_MATH=split UZ_DP_TABLES=0"; do
    echo "== NEW lib, $cfg: steps with 0 differing tensors (of 6): $(env $cfg bash -c "$(declare -f run); run")"
  done
  export UZ_LIB=$PWD/tools/tmp/libuz_hip_old.so
  for cfg in "UZ_DP_TABLES=1" "UZ_DIAG_BATCH=16" "UZ_DIAG_NETS=n,n UZ_DIAG_BATCH=16" "UZ_DIAG_NETS=n,n UZ_LANES=3"; do
    echo "== OLD lib, $cfg: steps with 0 differing tensors (of 6, stops at the first difference): $(env $cfg bash -c "$(declare -f run); run")"
  done
} > gpurun_out/r4_call92.txt 2>&1
