for t in "-" "pedersen_bundle=0" "pedersen_bundle_waves=2048" "pedersen_bundle_waves=512" "-" "pedersen_bundle=0"; do
  for tile in 16 17; do
    if [ "$t" = "-" ]; then unset ACVM_TUNING; else export ACVM_TUNING="$t"; fi
    echo -n "tile $tile: "; timeout 600 python tools/t_leg.py arith_pedersen 20 $tile 2>&1 | tail -1
  done
done
