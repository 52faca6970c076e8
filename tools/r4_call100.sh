# fp32-MFMA-only leg: does the one-launch BatchNorm (a workgroup needs a whole CU) cost more than it gives beside 2.5 ms launches?  lanes?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export UZ_CONV_MATH=f32
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="python bench.py --steps 20 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2; do
  $B 2>/dev/null | line "f32 default"
  UZ_BN_MID=0 $B 2>/dev/null | line "f32 UZ_BN_MID=0"
  UZ_BN_MID=0 UZ_BN_MID_FWD=0 $B 2>/dev/null | line "f32 UZ_BN_MID=0 UZ_BN_MID_FWD=0"
  UZ_LANES=3 $B 2>/dev/null | line "f32 UZ_LANES=3"
  UZ_LANES=3 UZ_BN_MID=0 UZ_BN_MID_FWD=0 $B 2>/dev/null | line "f32 UZ_LANES=3 UZ_BN_MID=0 UZ_BN_MID_FWD=0"
done
