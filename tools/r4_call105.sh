# What would fewer matrix products per output buy?  Timing-only experiment builds of the split kernels with 2 / 1 of the 3 piece products
# (csrc: UZ_EXP_PRODUCTS; staging, LDS images and epilogues unchanged; results lose accuracy but stay finite - real data, real clocks).
# Build the two experiment libraries first (they are not part of the product build and are not committed):
#   make -C unet-zoo_amd/csrc VARIANT=p2 XFLAGS=-DUZ_EXP_PRODUCTS=2 && make -C unet-zoo_amd/csrc VARIANT=p1 XFLAGS=-DUZ_EXP_PRODUCTS=1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/unet-zoo_amd
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for lib in libuz_hip.so libuz_hip_p2.so libuz_hip_p1.so; do
  echo "== $lib"
  for shape in "224 128 128 128" "128 128 128 128" "192 192 64 64" "192 192 32 32"; do
    echo "-- $shape"; UZ_LIB=$R/$lib python tools/bench_conv_packed.py $shape 32 10 0.5 2>&1 | tail -2
  done
done
for r in 1 2; do
  for lib in libuz_hip.so libuz_hip_p2.so libuz_hip_p1.so; do
    UZ_LIB=$R/$lib python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | line "step $lib"
  done
done
