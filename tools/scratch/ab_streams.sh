#!/bin/bash
# A/B of the look-ahead update on its own stream (GPMI_LA_STREAM) and of the two-stream many-right-hand-side solve
# (GPMI_TRSM_SPLIT): headline step time, alternating, plus the bit-identity of factor and predict across the variants.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
t=[k for k in r['kernels'] if 'trsm_rows' in k['kernel']]
print('ms %.3f  all_trailing %.2f  dominant %.2f  trsm %.2f TF/s %.3f ms' % (d['ms_per_step'], r['all_trailing']['achieved'], r['achieved'], t[0]['achieved'] if t else 0, t[0]['avg_ms'] if t else 0))"; }
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-sharded --no-cpu-baseline 2>gpurun_out/ab_err.txt | line || tail -5 gpurun_out/ab_err.txt; }
for rep in 1 2; do
  run GPMI_LA_STREAM=0 GPMI_TRSM_SPLIT=0
  run GPMI_LA_STREAM=1 GPMI_TRSM_SPLIT=0
  run GPMI_LA_STREAM=0 GPMI_TRSM_SPLIT=1
  run GPMI_LA_STREAM=1 GPMI_TRSM_SPLIT=1
done
timeout 900 python - <<'PY'
import os, subprocess, sys, json
code = r'''
import numpy as np, os, hashlib
from inference_amd.gp import GpRegressor, SquaredExponential
rng = np.random.default_rng(5)
N, d, M = 16384, 8, 1024
x = rng.uniform(0, 1, (N, d)); y = np.sin(x.sum(1)) + 0.1 * rng.normal(size=N)
gp = GpRegressor(x, y, y_err=np.full(N, 0.1), kernel=SquaredExponential(), hyperpars=np.array([0.0, 0.0] + [-0.5] * d))
mu, sig = gp(rng.uniform(0, 1, (M, d)))
Lf = gp.engine.get_L()
print(hashlib.sha256(Lf.tobytes()).hexdigest()[:16], hashlib.sha256(mu.tobytes()).hexdigest()[:16], hashlib.sha256(sig.tobytes()).hexdigest()[:16])
'''
out = {}
for la in "01":
    for sp in "01":
        env = dict(os.environ, GPMI_LA_STREAM=la, GPMI_TRSM_SPLIT=sp, PYTHONPATH="inference-tools_amd")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        out[la + sp] = r.stdout.strip().splitlines()[-1] if r.returncode == 0 else ("FAILED " + r.stderr[-400:])
        print("LA", la, "SPLIT", sp, out[la + sp])
print("bit-identical:", len(set(out.values())) == 1)
PY
