"""Error / warning texts of the GP classes.  The wording follows the reference so that callers that
match on messages keep working (regression.py:98-130, 226-234, 253-318, 331-348; optimisation.py:167-174);
keeping them here keeps the numerical classes readable."""


def framed(owner: str, kind: str, *lines: str) -> str:
    body = "".join(f"\n>> {line}" for line in lines)
    return f"\n\n[ {owner} {kind} ]{body}\n"


def framed_plain(owner: str, *lines: str) -> str:
    """The un-indented frame of inversion.py's messages."""
    body = "".join(f"\n>> {line}" for line in lines)
    return f"\n\n[ {owner} error ]{body}\n"


def y_not_1d(shape):
    return framed("GpRegressor", "error", f"'y' argument must be a 1D array, but instead has shape {shape}")


def x_not_2d(ndim, shape):
    return framed("GpRegressor", "Error", "'x' argument must be a 2D array, but instead has",
                  f"{ndim} dimensions and shape {shape}.")


def xy_mismatch(xshape, ysize):
    return framed("GpRegressor", "Error", "The first dimension of the 'x' array must be equal in size",
                  "to the 'y' array.", f"'x' has shape {xshape}, but 'y' has size {ysize}.")


def wrong_hyperpar_count(expected, given):
    return framed("GpRegressor", "error", "An incorrect number of hyper-parameter values were passed via the",
                  "'hyperpars' keyword argument:",
                  f"There are {expected} hyper-parameters but {given} values were given.")


def not_an_array(keyword, expected, given):
    return framed("GpRegressor", "error", f"The '{keyword}' keyword argument should be given as a numpy array:",
                  f"Expected type {expected} but type {given} was given.")


Y_COV_SHAPE = framed("GpRegressor", "error", "The 'y_cov' keyword argument was passed an array with an incorrect",
                     "shape. 'y_cov' must be a 2D array of shape (N,N), where 'N' is the",
                     "number of given y-data values.")
Y_COV_ASYMMETRIC = framed("GpRegressor", "error", "The covariance matrix passed to the 'y_cov' keyword argument",
                          "is not symmetric.")
Y_ERR_AND_Y_COV = framed("GpRegressor", "warning", "Only one of the 'y_err' and 'y_cov' keyword arguments should",
                         "be specified. Only the input to 'y_cov' will be used - the",
                         "input to 'y_err' will be ignored.")
Y_ERR_SHAPE = framed("GpRegressor", "error", "The 'y_err' keyword argument was passed an array with an",
                     "incorrect shape. 'y_err' must be a 1D array of length 'N',",
                     "where 'N' is the number of given y-data values.")
BAD_OPTIMIZER = (
    "\nAn invalid option was passed to the 'optimizer' keyword argument."
    "\nThe default option 'bfgs' was used instead."
    "\nValid options are 'bfgs' and 'diffev'.\n"
)


def points_not_2d(ndim, shape):
    return framed("GpRegressor", "error", "'points' argument must be a 2D array, but given array",
                  f"has {ndim} dimensions and shape {shape}.")


def points_wrong_width(n_dim, shape):
    return framed("GpRegressor", "error", "The second dimension of the 'points' array must have size",
                  "equal to the number of dimensions of the input data.",
                  f"The input data have {n_dim} dimensions but 'points' has shape {shape}.")


def no_device_kernel(cov_type):
    return framed("GpRegressor", "error", f"The covariance function {cov_type} has no MI355X device kernel.",
                  "Supported: SquaredExponential, RationalQuadratic, each optionally + WhiteNoise().")


NEW_Y_ERR_REQUIRED = framed("GpOptimiser", "error", "'new_y_err' argument of the 'add_evaluation' method must be",
                            "specified if the 'y_err' argument was specified when the",
                            "instance of GpOptimiser was initialised.")


def check_inverter_shapes(y, y_err, model_matrix, positions):
    """Shape checks of GpLinearInverter's constructor (inversion.py:63-113): same conditions, exception type
    and wording, in the order the reference applies them."""
    who = "GpLinearInverter"
    problems = (
        (model_matrix.ndim != 2, ("'model_matrix' argument must be a 2D numpy.ndarray",)),
        (y.ndim != y_err.ndim != 1 or y.size != y_err.size,
         ("'y' and 'y_err' arguments must be 1D numpy.ndarray", "of equal size.")),
        (model_matrix.ndim == 2 and model_matrix.shape[0] != y.size,
         ("The size of the first dimension of 'model_matrix' must", "equal the size of 'y', however they have shapes",
          f"{model_matrix.shape}, {y.shape}", "respectively.")),
        (positions.ndim != 2,
         ("'parameter_spatial_positions' must be a 2D numpy.ndarray, with the",
          "size of first dimension being equal to the number of model parameters",
          "and the size of the second dimension being equal to the number of", "spatial dimensions.")),
        (model_matrix.ndim == 2 and positions.ndim == 2 and model_matrix.shape[1] != positions.shape[0],
         ("The size of the second dimension of 'model_matrix' must be equal",
          "to the size of the first dimension of 'parameter_spatial_positions',", "however they have shapes",
          f"{model_matrix.shape}, {positions.shape}", "respectively.")),
    )
    for failed, lines in problems:
        if failed:
            raise ValueError(framed_plain(who, *lines))
