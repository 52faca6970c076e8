cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for r in 1 2 3; do
  $B 2>/dev/null | tail -1 | cut -c1-140
  UZ_NODE_PRIORITY=255 UZ_NODE_PRIORITY_VERBOSE=1 $B 2>gpurun_out/prio.err | tail -1 | cut -c1-140; grep "node prior" gpurun_out/prio.err | head -2
  UZ_NODE_PRIORITY=1024 $B 2>/dev/null | tail -1 | cut -c1-140
done
