cd $GRAFT_REPO_ROOT
python -m pytest tests/test_phiseg_gpu.py -x -q -p no:cacheprovider -s -k trajectory 2>&1 | grep -E "^step|passed|failed|Error" | head
for s in "192 192 32 32" "128 128 32 32" "192 192 16 16" "256 256 16 16" "320 192 32 32"; do echo "== $s"; python tools/bench_conv_packed.py $s 32 10 0.5 2>&1 | tail -2; python tools/stamp_conv.py $s 2>&1 | tail -3; done
for r in 1 2; do python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>&1 | tail -1 | cut -c1-140; done
