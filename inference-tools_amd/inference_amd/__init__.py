"""
inference_amd — MI355X-native drop-in for the Gaussian-process regression hot
path of C-bowman/inference-tools (`inference.gp`): same class surface
(`GpRegressor`, `GpOptimiser`, kernels, means, acquisition functions), with the
covariance build, Cholesky factorisation / solves and the log-marginal-likelihood
loop running as hand-written HIP kernels for gfx950 behind the C-ABI of
`include/gpmi.h` (bound through ctypes in `inference_amd._lib`).

There is no CPU fallback: importing works anywhere, but any computation raises
`GpmiUnavailable` when libgpmi.so or a GPU is missing.
"""
__version__ = "0.1.0"
