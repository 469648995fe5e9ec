# NOTE: GPMI_TRSM_LOOKAHEAD was a temporary patch of trsm_rows_forward (profiles/HISTORY.md R6.20), not in the library.
# A/B of GPMI_TRSM_LOOKAHEAD (the inverse factor's row solve with the next product beside the update) on ONE box
python3 - <<'PY'
import os, subprocess, sys
code = r'''
import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "inference-tools_amd")]
import numpy as np, workloads as wl
from inference_amd.gp import GpRegressor
out = {}
for n, d in ((8192, 8), (16384, 8), (4200, 3)):
    x, y, e = wl.synthetic_dataset(2, n, d)
    th = wl.timing_theta(wl.SE, y, d)
    gp = GpRegressor(x, y, y_err=e, hyperpars=th)
    gp.prepare_gradient()
    v, g = gp.marginal_likelihood_gradient(th)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); v, g = gp.marginal_likelihood_gradient(th); ts.append(time.perf_counter() - t0)
    lv, lg = gp.loo_likelihood_gradient(th)
    print(f"GPMI_TRSM_LOOKAHEAD={os.environ.get('GPMI_TRSM_LOOKAHEAD','1')} N={n}: LML + gradient {min(ts)*1e3:.2f} ms")
    out[f"v{n}"] = np.array([v, lv]); out[f"g{n}"] = np.concatenate([g, lg])
    gp.engine.close()
np.savez(sys.argv[1], **out)
'''
for v in ("0", "1", "0", "1"):
    subprocess.run([sys.executable, "-c", code, f"/tmp/tl{v}.npz"], env=dict(os.environ, GPMI_TRSM_LOOKAHEAD=v), check=True)
import numpy as np
a, b = np.load("/tmp/tl0.npz"), np.load("/tmp/tl1.npz")
print("bit-identical:", all(np.array_equal(a[k], b[k]) for k in a.files))
PY
