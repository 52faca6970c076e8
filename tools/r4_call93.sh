#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export UZ_DIAG_STEPS=5
run() { python tools/diag_dp_race.py 2>/dev/null | grep "^step.*params" | grep -c "differ: 0"; }
{ for seed in 11 12 13 14 15 16; do
    for cfg in "UZ_DIAG_NETS=n,n UZ_LANES=3" "UZ_DIAG_NETS=n,n" "UZ_DIAG_NETS=o,o"; do
      echo "seed $seed $cfg: identical steps (of 5): $(env UZ_DIAG_SEED=$seed $cfg bash -c "$(declare -f run); run")"
    done
  done
} > gpurun_out/r4_call93.txt 2>&1
