# Durations of the two triangular-sweep kernels in a fit + predict at N = 8192 and 16384 (rocprofv3 kernel stats),
# and a digest of the results (bit-identity across builds).   usage: bash tools/probes/sweep_stats.sh
mkdir -p gpurun_out/r06/sw
export TMPDIR=/tmp
for n in 8192 16384; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/sw/n$n -o t -- python3 tools/fit_digest.py /tmp/o$n.npz $n > /dev/null 2>&1
  f=$(find gpurun_out/r06/sw/n$n -name '*kernel_stats.csv' | head -1)
  echo "N=$n"
  [ -n "$f" ] && grep -i "trsv_\|Name" "$f" | cut -c1-200
done
timeout 100 python3 -c "
import numpy as np
for n in (8192,16384):
    z=np.load(f'/tmp/o{n}.npz'); print(n, float(z['logdet1'][0]).hex(), float(z['alpha1'].sum()).hex(), float(z['mu1'].sum()).hex())
"
