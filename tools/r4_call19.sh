cd $GRAFT_REPO_ROOT
for m in default split; do
  for cfg in "" "UZ_PACK_ACT=0 UZ_PACK_DY=0" "UZ_BN_MID_FWD=0" "UZ_BN_MID=0 UZ_BN_MID_FWD=0" "UZ_PACK_ACT=0 UZ_PACK_DY=0 UZ_BN_MID=0 UZ_BN_MID_FWD=0 UZ_WGRAD_TABLE=0 UZ_DBIAS_TABLE=0"; do
    echo "== math=$m cfg=[$cfg]"
    if [ $m = default ]; then env $cfg python tools/diag_digest.py phiseg_full_b32_digest 2>/dev/null | tail -1; else env UZ_CONV_MATH=split $cfg python tools/diag_digest.py phiseg_full_b32_digest 2>/dev/null | tail -1; fi
  done
done
