"""Stationary kernels (gpflow/kernels/stationaries.py:33-313).  K is built in one pass by
gpk_kernel_matrix; lengthscale scaling, the squared-distance expansion, the kernel profile and any
diagonal term are fused there."""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch

from ..base import Parameter, positive
from .. import ops
from .base import Kernel


class Stationary(Kernel):
    family = "SquaredExponential"

    def __init__(self, variance=1.0, lengthscales=1.0, **kwargs: Any):
        for kwarg in kwargs:
            if kwarg not in {"name", "active_dims"}:
                raise TypeError(f"Unknown keyword argument: {kwarg}")  # stationaries.py:56-58
        super().__init__(**kwargs)
        self.variance = Parameter(variance, transform=positive())
        self.lengthscales = Parameter(lengthscales, transform=positive())
        self._validate_ard_active_dims(self.lengthscales)

    def _validate_ard_active_dims(self, ard_parameter: Parameter) -> None:
        """gpflow/kernels/base.py:158-178"""
        if isinstance(self._active_dims, slice) or ard_parameter.numpy().ndim == 0:
            return
        if ard_parameter.numpy().shape[0] != len(self._active_dims):
            raise ValueError(
                f"Size of `active_dims` {self._active_dims} does not match size of ard parameter "
                f"({ard_parameter.numpy().shape[0]})")

    @property
    def ard(self) -> bool:
        return self.lengthscales.numpy().ndim > 0

    def K_diag(self, X) -> torch.Tensor:
        """stationaries.py:82-83: exactly sigma^2"""
        X = ops.to_device(X)
        return torch.full(X.shape[:-1], float(self.variance.numpy()), dtype=torch.float64, device=X.device)

    # device entry used by the covariance dispatchers and the fused model paths ------------------
    def hyper(self):
        """(family, variance, lengthscales) as host values for the C-ABI."""
        return self.family, float(self.variance.numpy()), np.asarray(self.lengthscales.numpy(), dtype=np.float64)

    def K_into(self, X, X2, out, *, diag_add: float = 0.0, lower_only: bool = False):
        family, var, ls = self.hyper()
        return ops.kernel_matrix(X, X2, variance=var, lengthscales=ls, family=family, diag_add=diag_add,
                                 lower_only=lower_only, out=out)


class IsotropicStationary(Stationary):
    def K(self, X, X2=None) -> torch.Tensor:
        """stationaries.py:103-105 (+ square_distance utilities/ops.py:105-122).  Leading batch
        dims are flattened to rows and restored: [batch..., N, D] x [batch2..., N2, D] ->
        [batch..., N, batch2..., N2] (or [batch..., N, N] when X2 is None)."""
        X = ops.to_device(X)
        D = X.shape[-1]
        if X2 is None:
            if X.dim() == 2:
                return self.K_into(X, None, None)
            lead = X.shape[:-2]
            Xf = X.reshape(-1, X.shape[-2], D)
            outs = [self.K_into(Xf[b].contiguous(), None, None) for b in range(Xf.shape[0])]
            return torch.stack(outs).reshape(*lead, X.shape[-2], X.shape[-2])
        X2 = ops.to_device(X2)
        Xf = X.reshape(-1, D).contiguous()
        X2f = X2.reshape(-1, D).contiguous()
        Kf = self.K_into(Xf, X2f, None)
        return Kf.reshape(*X.shape[:-1], *X2.shape[:-1])


class SquaredExponential(IsotropicStationary):
    """stationaries.py:194-210"""
    family = "SquaredExponential"


class Matern12(IsotropicStationary):
    """stationaries.py:244-255"""
    family = "Matern12"


class Matern32(IsotropicStationary):
    """stationaries.py:270-283"""
    family = "Matern32"


class Matern52(IsotropicStationary):
    """stationaries.py:298-313"""
    family = "Matern52"
