cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --timeout=1800 -p no:cacheprovider -x 2>&1 | tail -2
B="python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2; do $B 2>/dev/null | tail -1 | cut -c1-140; done
