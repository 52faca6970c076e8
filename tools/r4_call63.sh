cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2 3; do
  for c in 0 16 8 32; do echo -n "chunk $c: "; UZ_WGRAD_TABLE_CHUNK=$c $B 2>/dev/null | tail -1 | cut -c60-100; done
done
python -m pytest tests/test_split_storage_gpu.py tests/test_phiseg_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
