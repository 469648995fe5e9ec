from inference_amd.gp.regression import GpRegressor
from inference_amd.gp.mean import ConstantMean, LinearMean, QuadraticMean
from inference_amd.gp.covariance import (
    SquaredExponential,
    RationalQuadratic,
    WhiteNoise,
)

__all__ = [
    "GpRegressor",
    "ConstantMean",
    "LinearMean",
    "QuadraticMean",
    "SquaredExponential",
    "RationalQuadratic",
    "WhiteNoise",
]
