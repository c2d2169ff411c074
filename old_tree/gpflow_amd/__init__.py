"""gpflow_amd -- MI355X-native dense-GP hot path behind the GPflow model/posterior surface.

    import gpflow_amd as gpflow
    m = gpflow.models.GPR((X, Y), gpflow.kernels.RBF())
    m.log_marginal_likelihood()

Device work is hand-written HIP (gfx950) in libgpk.so behind the C-ABI of include/gpk.h; tensors are
fp64 torch tensors on the HIP device.  There is no CPU fallback.
"""
from . import config  # noqa: F401,E402
from .base import Module, Parameter, set_trainable  # noqa: F401
from .config import default_float, default_int, default_jitter  # noqa: F401
from . import (conditionals, covariances, functions, inducing_variables, kernels, kullback_leiblers,  # noqa: F401
               likelihoods, logdensities, mean_functions, models, optimizers, posteriors, priors, training, utilities)

__version__ = "0.1.0"
