#!/bin/bash
# usage: tools/build_gemm_variant.sh <name> <extra hipcc flags...>  ->  inference-tools_amd/inference_amd/lib/libgpmi_<name>.so
# (like build_variant.sh, for flags that change gemm_tiles.h: every translation unit that includes it is rebuilt)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$ROOT/tools/variants; mkdir -p $out/obj_$name
cd $ROOT/inference-tools_amd/csrc
for f in api api_regression api_mix api_linv api_dense kbuild gemm_f64 potrf potrf_flow solve grad predgrad mix comm; do
  if grep -q "gemm_tiles.h" $f.hip || [ ! -f build/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I../../include -I. "$@" -c $f.hip -o $out/obj_$name/$f.o
  else
    cp build/$f.o $out/obj_$name/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/inference-tools_amd/inference_amd/lib/libgpmi_$name.so $out/obj_$name/*.o -ldl
echo libgpmi_$name.so
