# persistent-workgroup experiment on the 64-channel-tile split convolutions (UZ_CONV_PERSIST=n: launches of more than n workgroups run as n persistent ones)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/unet-zoo_amd/libuz_hip_persist.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | line "main"
  for n in 0 248 240 224 192; do
    UZ_LIB=$V UZ_CONV_PERSIST=$n python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | line "variant persist=$n"
  done
done
UZ_LIB=$V UZ_CONV_PERSIST=240 python -m pytest tests/test_phiseg_gpu.py -m gpu -q -x -p no:cacheprovider -k "digest or golden" 2>&1 | tail -3
