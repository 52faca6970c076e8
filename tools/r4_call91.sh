#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export UZ_DP_TABLES=0 UZ_DIAG_STEPS=6 UZ_DIAG_NETS=o,o
run() { python tools/diag_dp_race.py 2>/dev/null | grep "^step.*params" | cut -c1-100 | tr '\n' ';'; echo; }
{ echo "== NEW lib, DP no tables, o,o"; run
  echo "== NEW lib, DP tables (default), o,o"; UZ_DP_TABLES=1 run
  export UZ_LIB=$PWD/tools/tmp/libuz_hip_old.so
  echo "== old lib baseline"; run
  echo "== old lib, UZ_DECOUPLE_WGRAD=0"; UZ_DECOUPLE_WGRAD=0 run
  echo "== old lib, UZ_BN_FOLD_REDUCE=0"; UZ_BN_FOLD_REDUCE=0 run
  echo "== old lib, UZ_BN_MID=0"; UZ_BN_MID=0 run
  echo "== old lib, UZ_PREPACK=0"; UZ_PREPACK=0 run
  echo "== old lib, UZ_CONV_MATH=f32"; UZ_CONV_MATH=f32 run
  echo "== old lib, UZ_BN_FUSE_STATS=0"; UZ_BN_FUSE_STATS=0 run
} > gpurun_out/r4_call91.txt 2>&1
