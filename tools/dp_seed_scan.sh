cd $GRAFT_REPO_ROOT
for s in 100 300 500 700 900; do
  UZ_DP_TEST_SEED=$s HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tests/dp_gpu_worker.py 2>&1 | grep "rank 0" | tr '\n' ' '; echo
done
