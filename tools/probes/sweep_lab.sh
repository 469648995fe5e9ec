# build (here or on the GPU box) and run tools/probes/sweep_lab.hip:  bash tools/probes/sweep_lab.sh [n ...]
set -e
cd "$(dirname "$0")/../.."
C=inference-tools_amd/csrc
objs=$(ls $C/build/*.o | grep -v '/solve.o')
mkdir -p gpurun_out/probes
if [ ! -x gpurun_out/probes/sweep_lab ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I$C -c tools/probes/sweep_lab.hip -o gpurun_out/probes/sweep_lab.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 gpurun_out/probes/sweep_lab.o $objs -ldl -o gpurun_out/probes/sweep_lab
fi
[ -n "$BUILD_ONLY" ] || timeout 120 gpurun_out/probes/sweep_lab "$@"
