#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('ms %.3f  all_trailing %.2f  dominant %.2f' % (d['ms_per_step'], r['all_trailing']['achieved'], r['achieved']))"; }
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-sharded --no-cpu-baseline 2>gpurun_out/ab_err.txt | line || tail -5 gpurun_out/ab_err.txt; }
run GPMI_LA_MERGE=0
run GPMI_LA_MERGE=1
run GPMI_LA_MERGE=1 GPMI_SLICE_LA_ADJ=64
run GPMI_LA_MERGE=1 GPMI_SLICE_LA_ADJ=128
run GPMI_LA_MERGE=1 GPMI_SLICE_LA_ADJ=256
run GPMI_LA_MERGE=0
run GPMI_LA_MERGE=1
timeout 900 python - <<'PY'
import os, subprocess, sys
code = r'''
import numpy as np, hashlib
from inference_amd.gp import GpRegressor, SquaredExponential
rng = np.random.default_rng(5)
N, d, M = 16384, 8, 256
x = rng.uniform(0, 1, (N, d)); y = np.sin(x.sum(1)) + 0.1 * rng.normal(size=N)
gp = GpRegressor(x, y, y_err=np.full(N, 0.1), kernel=SquaredExponential(), hyperpars=np.array([0.0, 0.0] + [-0.5] * d))
mu, sig = gp(rng.uniform(0, 1, (M, d)))
Lf = gp.engine.get_L()
print(hashlib.sha256(Lf.tobytes()).hexdigest()[:16], hashlib.sha256(mu.tobytes()).hexdigest()[:16], hashlib.sha256(sig.tobytes()).hexdigest()[:16])
'''
out = {}
for m in "01":
    env = dict(os.environ, GPMI_LA_MERGE=m, PYTHONPATH="inference-tools_amd")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    out[m] = r.stdout.strip().splitlines()[-1] if r.returncode == 0 else ("FAILED " + r.stderr[-600:])
    print("MERGE", m, out[m])
print("bit-identical:", len(set(out.values())) == 1)
PY
