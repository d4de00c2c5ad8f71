cd $GRAFT_REPO_ROOT
for cb in 0 64 128 256; do for kb in 38 44 52; do
  r=$(DTC_RA_TILE_CHBLOCK=$cb DTC_RA_TILE_LDS16_KB=$kb python tools/r06/ab_fused16.py --mode contract --iters 30 2>/dev/null | tail -1)
  echo "chblock=$cb lds16=$kb : $r"
done; done
