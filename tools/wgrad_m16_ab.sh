#!/bin/bash
# A/B of the MFMA shape in the weight gradient (UZ_WG_M16=0: 32x32x16, 1: 16x16x32), isolated launches, both grid targets
mkdir -p gpurun_out; out=gpurun_out/wgrad_m16_ab.txt; : > $out
for shape in "224 128 128 128" "128 128 128 128" "256 192 64 64" "192 192 64 64" "320 192 32 32" "192 192 32 32"; do
  for tgt in 128 256; do
    for m in 0 1 0 1; do
      echo "== $shape target $tgt M16=$m" >> $out
      env UZ_WG9=0 UZ_WG_M16=$m UZ_WGS_TARGET=$tgt timeout 300 python tools/bench_conv_packed.py $shape 32 20 2>&1 | grep -E "^(fp32|packed)" >> $out
    done
  done
done
