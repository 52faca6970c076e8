# fragment reads one tap ahead of the MFMAs in the 64-channel-tile kernels too (experiment build: make -C unet-zoo_amd/csrc VARIANT=pref XFLAGS=-DUZ_EXP_PREF_ALL=1)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT/unet-zoo_amd
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
{
for lib in libuz_hip.so libuz_hip_pref.so; do
  echo "== $lib"
  for shape in "224 128 128 128" "128 128 128 128" "192 192 64 64"; do
    echo "-- $shape"; UZ_LIB=$R/$lib python tools/bench_conv_packed.py $shape 32 10 0.5 2>&1 | tail -2
  done
done
for r in 1 2 3; do
  for lib in libuz_hip.so libuz_hip_pref.so; do
    UZ_LIB=$R/$lib python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | line "step $lib"
  done
done
UZ_LIB=$R/libuz_hip_pref.so python -m pytest tests/test_ops_gpu.py tests/test_full_configs_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv or split" 2>&1 | tail -3
} > gpurun_out/r4_call109.txt 2>&1
tail -40 gpurun_out/r4_call109.txt
