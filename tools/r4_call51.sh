cd $GRAFT_REPO_ROOT
python -m pytest tests/test_b16_storage_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300
for i in 1 2; do python bench.py --model phiseg3d --steps 20 --warmup 5 --skip-cpu --no-profile 2>/dev/null | tail -1 | cut -c1-200; done
python -m pytest tests/test_phiseg3d.py -m gpu -q -p no:cacheprovider -k "bf16" 2>&1 | tail -2
