"""Phase cycle counts of potrf_diag (debug): factor a random SPD 128 x 128 block through gpmi_dev_potrf."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inference-tools_amd"))
import numpy as np
from inference_amd import _lib
os.environ["GPMI_DIAG_STAMPS"] = "1"
h = _lib.Handle(0)
n, ld = 128, 160
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n)); Am = B @ B.T + n * np.eye(n)
buf = np.zeros((n, ld)); buf[:, :n] = Am
d = C.c_void_p(); h.call("gpmi_dev_alloc", buf.nbytes, C.byref(d))
for _ in range(3):
    h.call("gpmi_dev_upload", d, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    info = C.c_int(); h.call("gpmi_dev_potrf", d, n, ld, C.byref(info))
out = np.empty_like(buf); h.call("gpmi_dev_download", out.ctypes.data_as(C.c_void_p), d, buf.nbytes)
L = np.tril(out[:, :n]); print("max err", np.abs(L - np.linalg.cholesky(Am)).max())
