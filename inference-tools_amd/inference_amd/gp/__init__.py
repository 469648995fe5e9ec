from inference_amd.gp.regression import GpRegressor
from inference_amd.gp.optimisation import GpOptimiser
from inference_amd.gp.acquisition import (
    ExpectedImprovement,
    UpperConfidenceBound,
    MaxVariance,
)
from inference_amd.gp.mean import ConstantMean, LinearMean, QuadraticMean
from inference_amd.gp.covariance import (
    SquaredExponential,
    RationalQuadratic,
    WhiteNoise,
)

__all__ = [
    "GpRegressor",
    "GpOptimiser",
    "ExpectedImprovement",
    "UpperConfidenceBound",
    "MaxVariance",
    "ConstantMean",
    "LinearMean",
    "QuadraticMean",
    "SquaredExponential",
    "RationalQuadratic",
    "WhiteNoise",
]
