cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for e in 0 1 0 1; do echo "== UZ_EXP=$e"; for l in "224 128 128 128" "128 128 128 128" "256 192 64 64"; do UZ_EXP=$e python tools/bench_conv.py $l 2>&1 | tail -3 | tr '\n' ' '; echo; done; done
for e in 0 1 0 1; do UZ_EXP=$e python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | cut -c90-175; done
