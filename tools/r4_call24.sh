cd $GRAFT_REPO_ROOT
run() { echo "== [$1]"; env $1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tests/dp_gpu_worker.py 2>&1 | grep "rank 0:"; }
run "UZ_X=0"
run "UZ_BN_MID_FWD=0"
run "UZ_BN_MID=0"
run "UZ_PACK_ACT=0 UZ_PACK_DY=0"
run "UZ_BN_FOLD_DGRAD=0"
run "UZ_BN_MID_FWD=0 UZ_BN_MID=0 UZ_PACK_ACT=0 UZ_PACK_DY=0 UZ_BN_FOLD_DGRAD=0"
timeout 600 python -m pytest tests/test_split_storage_gpu.py tests/test_ops_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
for r in 1 2 3; do
python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | cut -c1-140
UZ_BN_FOLD_DGRAD=0 python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | cut -c1-140
done
