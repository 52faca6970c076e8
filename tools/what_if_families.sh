# In-situ cost of every op family and of every resolution level: the PHiSeg step with that class of ops SKIPPED (diagnostic build, results garbage;
# tools/what_if.sh explains the build).  Usage on a GPU box: make -C unet-zoo_amd/csrc VARIANT=diag XFLAGS=-DUZ_DIAG; bash tools/what_if_families.sh
cd $GRAFT_REPO_ROOT
export UZ_LIB=$PWD/unet-zoo_amd/libuz_hip_diag.so
run() { echo "$1 : $(UZ_DIAG_SKIP="$1" python bench.py --allow-experiment --steps 30 --warmup 5 --skip-cpu --no-profile --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['config']['schedule']['tape_us']; print(d['ms_per_step'], 'ms', d['value'], 'images/s | tapes us: fwd', t['fwd'][-1], 'bwd', t['bwd'][-1])")"; }
run "none:0"
for f in wgrad:128 dgrad:128 fwd:128 bnf:128 bnb:128 bn:128 resample:128 conv:128; do run "$f"; done
for l in 128 64 32 16; do run "only:$l,conv:$l"; run "only:$l,conv:$l,bn:$l,resample:$l"; done
# the deep levels (NOTES_r6 section 5)
for f in wgrad:8 wgrad:16 "bnf:16,fwd:16" "bnb:16,dgrad:16,wgrad:16" "bn:16,conv:16,resample:8" "bn:8,conv:8,resample:4"; do run "$f"; done
run "none:0"
