cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg"
for q in "2 2" "3 3" "4 4"; do set -- $q; echo "== queues $1 $2"; GPU_MAX_HW_QUEUES=$1 DEBUG_HIP_FORCE_GRAPH_QUEUES=$2 $B 2>/dev/null | cut -c90-160; done
for l in 1 2; do echo "== phiseg lanes $l"; UZ_LANES=$l $B 2>/dev/null | cut -c90-160; done
for l in 1 2 3; do echo "== probunet lanes $l"; UZ_LANES=$l $B --model probunet 2>/dev/null | cut -c100-170; done
for l in 1 2; do echo "== unet lanes $l"; UZ_LANES=$l $B --model unet 2>/dev/null | cut -c90-160; done
for l in 1 2 3; do echo "== phiseg3d lanes $l"; UZ_LANES=$l $B --model phiseg3d --steps 10 2>/dev/null | cut -c100-190; done
