# A/B of GPMI_EARLY_FILL (residual + sentinel fills enqueued where the lane's stream idles) on ONE box
for rep in 1 2 3; do
  for v in 0 1; do echo -n "GPMI_EARLY_FILL=$v: "; GPMI_EARLY_FILL=$v python3 tools/fit_timeline.py 8192 60; done
done
for rep in 1 2; do
  for v in 0 1; do echo -n "GPMI_EARLY_FILL=$v: "; GPMI_EARLY_FILL=$v python3 tools/fit_timeline.py 16384 12; done
done
for v in 0 1; do GPMI_EARLY_FILL=$v python3 tools/fit_digest.py /tmp/ef$v.npz 16384; GPMI_EARLY_FILL=$v python3 tools/fit_digest.py /tmp/eg$v.npz 8192; done
python3 - <<'PY'
import numpy as np
for f in ('ef', 'eg'):
    a, b = np.load(f'/tmp/{f}0.npz'), np.load(f'/tmp/{f}1.npz')
    print(f, 'bit-identical:', all(np.array_equal(a[k], b[k]) for k in a.files))
PY
