# A/B of GPMI_BACKWARD_OB (the backward row solve of the spatial derivatives over 512-wide blocks with the inverse blocks
# instead of 128-wide steps) on ONE box: config 4's timings, and the values against each other
for rep in 1 2 3; do
  for v in 0 1; do echo -n "GPMI_BACKWARD_OB=$v: "; GPMI_BACKWARD_OB=$v python3 tools/config_bench.py cfg4 2>&1 | grep -i "EI" | head -1; done
done
for v in 0 1; do echo -n "GPMI_BACKWARD_OB=$v: "; GPMI_BACKWARD_OB=$v python3 tools/propose_bench.py 4096 32 2>&1 | head -c 300; echo; done
python3 - <<'PY'
import os, subprocess, sys, json
code = r'''
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "inference-tools_amd")]
import numpy as np, workloads as wl
from inference_amd.gp import GpRegressor
for n, d, m in ((4096, 4, 1000), (2500, 3, 77), (8192, 8, 300)):
    x, y, e = wl.synthetic_dataset(4, n, d)
    gp = GpRegressor(x, y, y_err=e, hyperpars=wl.timing_theta(wl.SE, y, d))
    dm, dv = gp.spatial_derivatives(wl.query_points(4, m, d))
    np.save(f"/tmp/sd_{os.environ['GPMI_BACKWARD_OB']}_{n}.npy", np.concatenate([dm.ravel(), dv.ravel()]))
'''
for v in ("0", "1"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GPMI_BACKWARD_OB=v), check=True)
import numpy as np
for n in (4096, 2500, 8192):
    a, b = np.load(f"/tmp/sd_0_{n}.npy"), np.load(f"/tmp/sd_1_{n}.npy")
    print(f"N={n}: max |new - old| / max |old| = {np.abs(a - b).max() / np.abs(a).max():.2e}")
PY
