"""Kernels on the hot path (gpflow/kernels): stationary family built by the HIP covariance builder, their sums and
products (`+` / `*`, kernels/base.py:216-329), plus the multi-output wrappers SharedIndependent / SeparateIndependent."""
from .base import Combination, Kernel, Product, Sum
from .stationaries import (Stationary, IsotropicStationary, SquaredExponential, Matern12, Matern32,
                           Matern52)
from .multioutput import MultioutputKernel, SharedIndependent, SeparateIndependent

RBF = SquaredExponential  # gpflow/kernels/__init__.py:50

__all__ = ["Kernel", "Combination", "Sum", "Product", "Stationary", "IsotropicStationary", "SquaredExponential", "RBF", "Matern12",
           "Matern32", "Matern52", "MultioutputKernel", "SharedIndependent", "SeparateIndependent"]
