#!/bin/bash
# A/B of the two weight-gradient forms (UZ_WG9=0: eight waves, tap groups; 1: four waves, nine taps per wave, two workgroups per CU)
# on the headline plan's shapes, at the two grid targets the models use.  Output: gpurun_out/wgrad9_ab.txt
mkdir -p gpurun_out; out=gpurun_out/wgrad9_ab.txt; : > $out
for shape in "224 128 128 128" "128 128 128 128" "256 192 64 64" "192 192 64 64" "320 192 32 32" "192 192 32 32" "256 256 16 16" "96 64 128 128"; do
  for tgt in 128 256; do
    for wg9 in 0 1; do
      echo "== $shape target $tgt UZ_WG9=$wg9 ${EXTRA}" >> $out
      env UZ_WG9=$wg9 UZ_WGS_TARGET=$tgt $EXTRA timeout 300 python tools/bench_conv_packed.py $shape 32 10 2>&1 | grep -v Warn >> $out
    done
  done
done
cat $out
