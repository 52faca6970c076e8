cd $GRAFT_REPO_ROOT
python tools/op_profile.py 32 phiseg > gpurun_out/r4_op_profile_new.txt 2>&1
UZ_PACK_ACT=0 UZ_PACK_DY=0 UZ_FOLD_BN_BWD=0 python tools/op_profile.py 32 phiseg > gpurun_out/r4_op_profile_off.txt 2>&1
for r in 1 2 3; do
python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
UZ_PACK_ACT=0 UZ_PACK_DY=0 UZ_FOLD_BN_BWD=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
done
