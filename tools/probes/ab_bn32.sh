# A/B of GPMI_CHAIN_BN32 (32 x 32 mirrored tiles for the products with the 512 x 512 inverse blocks) on ONE box
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== GPMI_CHAIN_BN32=$v"
    GPMI_CHAIN_BN32=$v python3 tools/predict_time.py 16384 8192 2>&1 | cut -c1-230
    GPMI_CHAIN_BN32=$v python3 tools/config_bench.py cfg4 2>&1 | grep -i "EI" | head -2
  done
done
