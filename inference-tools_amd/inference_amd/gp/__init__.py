"""Device-backed Gaussian-process classes; the public names are those of `inference.gp`
(reference inference/gp/__init__.py)."""
from inference_amd.gp.covariance import ChangePoint, HeteroscedasticNoise, RationalQuadratic, SquaredExponential, WhiteNoise
from inference_amd.gp.mean import ConstantMean, LinearMean, QuadraticMean
from inference_amd.gp.acquisition import ExpectedImprovement, MaxVariance, UpperConfidenceBound
from inference_amd.gp.regression import GpRegressor
from inference_amd.gp.optimisation import GpOptimiser
from inference_amd.gp.inversion import GpLinearInverter

__all__ = sorted(
    name for name, obj in list(globals().items()) if isinstance(obj, type) and not name.startswith("_")
)
