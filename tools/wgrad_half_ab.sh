#!/bin/bash
# A/B of the half-workgroup weight gradient (UZ_WG_HALF=1: 256 threads, 64 x 32 tile, two workgroups per CU) against the first form, isolated launches with split-storage operands
mkdir -p gpurun_out; out=gpurun_out/wgrad_half_ab.txt; : > $out
for shape in "224 128 128 128" "128 128 128 128" "256 192 64 64" "192 192 64 64" "320 192 32 32" "192 192 32 32"; do
  for tgt in 128 256; do
    for m in 0 1 0 1; do
      echo "== $shape target $tgt HALF=$m" >> $out
      env UZ_WG_HALF=$m UZ_WGS_TARGET=$tgt timeout 300 python tools/bench_conv_packed.py $shape 32 20 2>&1 | grep -E "^(packed)" >> $out
    done
  done
done
