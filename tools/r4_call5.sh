cd $GRAFT_REPO_ROOT
for s in "224 128 128 128" "128 128 128 128" "192 192 64 64" "192 192 32 32" "192 192 16 16" "32 32 128 128"; do echo "== $s"; python tools/bench_conv_packed.py $s 32 10 0.5 2>&1 | tail -2; done
echo "== 224 128 dense data"; python tools/bench_conv_packed.py 224 128 128 128 32 10 0.0 2>&1 | tail -2
python -m pytest tests/test_split_storage_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
python tools/op_profile.py 32 phiseg > gpurun_out/r4_op_profile_new2.txt 2>&1
for r in 1 2; do
python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
UZ_PACK_ACT=0 UZ_PACK_DY=0 UZ_FOLD_BN_BWD=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
UZ_FOLD_BN_BWD=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140
done
