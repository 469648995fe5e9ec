#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>   ->  gpurun_out/lib/libgpmi_<name>.so  (A/B builds; GPMI_LIB selects one)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
out=$ROOT/tools/variants; mkdir -p $out/obj_$name
cd $ROOT/inference-tools_amd/csrc
for f in api api_regression api_mix api_linv api_dense kbuild gemm_f64 potrf potrf_flow solve grad predgrad mix comm; do
  if [ "$f" = potrf ] || [ "$f" = potrf_flow ] || [ ! -f build/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I../../include -I. "$@" -c $f.hip -o $out/obj_$name/$f.o
  else
    cp build/$f.o $out/obj_$name/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libgpmi_$name.so $out/obj_$name/*.o -ldl
echo $out/libgpmi_$name.so
