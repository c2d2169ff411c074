"""Gaussian likelihood (gpflow/likelihoods/scalar_continuous.py:41-148) -- the conjugate case that
keeps the whole ELBO on the dense path."""
from __future__ import annotations

from math import sqrt
from typing import Optional

import numpy as np
import torch

from . import config, ops
from .base import Module, Parameter, positive
from .functions import Function
from .logdensities import gaussian

LOG2PI = float(np.log(2 * np.pi))


class Likelihood(Module):
    pass


class Gaussian(Likelihood):
    """Gaussian(variance=None, *, scale=None, variance_lower_bound=None), scalar_continuous.py:41-148.  `variance` / `scale` is a
    constant (a positive Parameter) or a Function of the inputs (gpflow/functions.py) -- a heteroskedastic likelihood whose value is
    clipped from below at the lower bound when it is evaluated (utilities/parameter_or_function.py:45-58)."""

    def __init__(self, variance=None, *, scale=None, variance_lower_bound: Optional[float] = None):
        self.variance_lower_bound = (config.default_likelihood_positive_minimum()
                                     if variance_lower_bound is None else float(variance_lower_bound))
        self.scale_lower_bound = sqrt(self.variance_lower_bound)
        if scale is None:
            if variance is None:
                variance = 1.0
            self.variance = self._prepare(variance, self.variance_lower_bound)
            self.scale = None
        else:
            assert variance is None, "Cannot set both `variance` and `scale`."
            self.variance = None
            self.scale = self._prepare(scale, self.scale_lower_bound)

    @staticmethod
    def _prepare(value, lower_bound: float):
        """prepare_parameter_or_function (utilities/parameter_or_function.py:27-39)"""
        if isinstance(value, Function):
            return value
        # (a Parameter handed in is re-wrapped like any other value, as the reference does: the NEW Parameter carries the lower-bound
        #  transform -- an identity- or otherwise-transformed one would let the optimiser take the variance below the bound --
        #  and inherits prior / trainable, base.py:155-161)
        return Parameter(value, transform=positive(lower=lower_bound))

    @property
    def is_heteroskedastic(self) -> bool:
        """The noise depends on the inputs (variance / scale is a Function)."""
        return isinstance(self.variance if self.variance is not None else self.scale, Function)

    @property
    def has_variance_parameter(self) -> bool:
        """Constant noise held as a `variance` Parameter -- what the hand-written reverse passes differentiate."""
        return isinstance(self.variance, Parameter)

    def noise_variance(self) -> float:
        """scalar_continuous.py:92-105 (constant-variance case)"""
        if self.is_heteroskedastic:
            raise ValueError("the noise variance of this likelihood depends on the inputs: use noise_for(X) / variance_at(X)")
        if self.variance is not None:
            return float(self.variance.numpy())
        return float(self.scale.numpy()) ** 2

    def _variance(self, X):
        """scalar_continuous.py:92-105: a float (constant) or a device tensor [..., N, Q] (Function, clipped at the lower bound)."""
        if not self.is_heteroskedastic:
            return self.noise_variance()
        if self.variance is not None:
            return torch.clamp(self.variance(ops.to_device(X)), min=self.variance_lower_bound)
        return torch.clamp(self.scale(ops.to_device(X)), min=self.scale_lower_bound) ** 2

    def noise_for(self, X):
        """What the device entry points take as `noise_variance`: the constant as a float, or one variance per row of X as a
        device tensor [N] -- variance_at(X) squeezed (model_utils.py:46-50, sgpr.py:207)."""
        if not self.is_heteroskedastic:
            return self.noise_variance()
        X = ops.to_device(X)
        if X.dim() != 2:
            raise ValueError("noise_for expects X [N, D]")
        return self.variance_at(X)[:, 0].contiguous()

    def noise_param_grads(self, X, g_noise):
        """[(Parameter, dF/d(constrained value))] of the noise function's parameters, given g_noise [N] = dF/d sigma_n^2 at the rows of
        X (what the per-row reverse passes return as "noise_variance") -- the chain rule through _variance (clip at the lower
        bound: no gradient where the function sits below it; scale: d s^2 = 2 s ds) and the Function itself."""
        X = ops.to_device(X)
        fn = self.variance if self.variance is not None else self.scale
        raw = fn(X)
        g = g_noise.reshape(-1, 1)
        if raw.shape[-1] != 1:
            raise NotImplementedError("gradients of a heteroskedastic noise function with more than one output column")
        if self.variance is not None:
            gbar = g * (raw > self.variance_lower_bound)
        else:
            gbar = g * (2.0 * torch.clamp(raw, min=self.scale_lower_bound)) * (raw > self.scale_lower_bound)
        return fn.backward(X, gbar)

    def variance_at(self, X) -> torch.Tensor:
        """scalar_continuous.py:107-111: [..., N, 1]"""
        X = ops.to_device(X)
        v = self._variance(X)
        if not torch.is_tensor(v):
            return torch.full(X.shape[:-1] + (1,), v, dtype=torch.float64, device=X.device)
        return torch.broadcast_to(v, X.shape[:-1] + (1,))

    def log_prob(self, X, F, Y):
        return gaussian(ops.to_device(Y), ops.to_device(F), self._variance(X)).sum(-1)

    def predict_mean_and_var(self, X, Fmu, Fvar):
        """scalar_continuous.py:127-130"""
        return Fmu.clone(), Fvar + self._variance(X)

    def predict_log_density(self, X, Fmu, Fvar, Y):
        """scalar_continuous.py:132-136"""
        return gaussian(ops.to_device(Y), Fmu, Fvar + self._variance(X)).sum(-1)

    def variational_expectations(self, X, Fmu, Fvar, Y) -> torch.Tensor:
        """scalar_continuous.py:139-148 -- per-row values [N] (elementwise glue; the summed form used
        by SVGP.elbo runs in gpk_gaussian_varexp_sum)."""
        v = self._variance(X)
        Y = ops.to_device(Y)
        logv = torch.log(v) if torch.is_tensor(v) else float(np.log(v))
        return (-0.5 * LOG2PI - 0.5 * logv - 0.5 * ((Y - Fmu) ** 2 + Fvar) / v).sum(-1)
