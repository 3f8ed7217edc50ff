cd $GRAFT_REPO_ROOT
for cfg in "" "RAL_UNET_BWD_WP=4" "RAL_UNET_BWD_WP=1" "RAL_UNET_BWD_WP=4 RAL_UNET_BWD_GRID=256" "RAL_UNET_BWD_WP=2 RAL_UNET_BWD_GRID=1024" "RAL_UNET_FWD_GRID=1024" "RAL_UNET_NREP=8" "RAL_UNET_NREP=4"; do
  echo "== $cfg"
  env $cfg python3 tools/unet_bench.py 2>&1 | grep -o '"train_ms": [0-9.]*\|"fwd_train_ms": [0-9.]*' | tr '\n' ' '; echo
done
