# A/B of library variants (tools/build_gemm_variant.sh <name> -D...) on the small-tile chain products: predict and config 4
# usage: bash tools/probes/ab_pf.sh <name> ...
D=$PWD/inference-tools_amd/inference_amd/lib
for rep in 1 2; do
  for lib in libgpmi.so "$@"; do
    [ "$lib" = libgpmi.so ] || lib=libgpmi_$lib.so
    echo "== $lib"
    GPMI_LIB=$D/$lib python3 tools/predict_time.py 16384 8192 2>&1 | cut -c1-230
    GPMI_LIB=$D/$lib python3 tools/config_bench.py cfg4 2>&1 | grep -i "ei\b\|ei \|EI" | head -4
  done
done
