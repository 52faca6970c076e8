cd $GRAFT_REPO_ROOT
run() { echo "== [$1]"; env $1 python tools/diag_midfwd.py 2>&1 | grep -E "^(flags|worst|loss)" ; }
run "UZ_X=0"
run "UZ_CONV_MATH=f32"
run "UZ_LANES=1"
run "UZ_BN_MID_FWD=0"
run "UZ_BN_MID_FWD=0 UZ_BN_FUSE_STATS=0"
run "UZ_CONV_MATH=f32 UZ_BN_MID_FWD=0"
run "UZ_CONV_MATH=split"
run "UZ_CONV_MATH=split UZ_BN_MID_FWD=0"
