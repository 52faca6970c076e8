#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export UZ_LIB=$PWD/tools/tmp/libuz_hip_old.so UZ_DP_TABLES=0 UZ_DIAG_STEPS=3
{ for k in "o,o" "n,n" "n,s" "s,o"; do echo "== nets $k"; UZ_DIAG_NETS=$k python tools/diag_dp_race.py 2>/dev/null | grep "^step.*params\|^    post" | cut -c1-170 | head -5; done
  echo "== nets n,o with UZ_LANES=1"; UZ_LANES=1 UZ_DIAG_NETS=n,o python tools/diag_dp_race.py 2>/dev/null | grep "^step.*params" | cut -c1-150
} > gpurun_out/r4_call90.txt 2>&1
