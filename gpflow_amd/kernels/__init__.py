"""Kernels on the hot path (gpflow/kernels): stationary family built by the HIP covariance builder,
plus the multi-output wrappers SharedIndependent / SeparateIndependent."""
from .base import Kernel
from .stationaries import (Stationary, IsotropicStationary, SquaredExponential, Matern12, Matern32,
                           Matern52)
from .multioutput import MultioutputKernel, SharedIndependent, SeparateIndependent

RBF = SquaredExponential  # gpflow/kernels/__init__.py:50

__all__ = ["Kernel", "Stationary", "IsotropicStationary", "SquaredExponential", "RBF", "Matern12",
           "Matern32", "Matern52", "MultioutputKernel", "SharedIndependent", "SeparateIndependent"]
