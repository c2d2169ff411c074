"""Gaussian likelihood (gpflow/likelihoods/scalar_continuous.py:41-148) -- the conjugate case that
keeps the whole ELBO on the dense path."""
from __future__ import annotations

from math import sqrt
from typing import Optional

import numpy as np
import torch

from . import config, ops
from .base import Module, Parameter, positive
from .logdensities import gaussian

LOG2PI = float(np.log(2 * np.pi))


class Likelihood(Module):
    pass


class Gaussian(Likelihood):
    def __init__(self, variance=None, *, scale=None, variance_lower_bound: Optional[float] = None):
        self.variance_lower_bound = (config.default_likelihood_positive_minimum()
                                     if variance_lower_bound is None else float(variance_lower_bound))
        self.scale_lower_bound = sqrt(self.variance_lower_bound)
        if scale is None:
            if variance is None:
                variance = 1.0
            self.variance = Parameter(variance, transform=positive(lower=self.variance_lower_bound))
            self.scale = None
        else:
            assert variance is None, "Cannot set both `variance` and `scale`."
            self.variance = None
            self.scale = Parameter(scale, transform=positive(lower=self.scale_lower_bound))

    def noise_variance(self) -> float:
        """scalar_continuous.py:92-105 (constant-variance case)"""
        if self.variance is not None:
            return float(self.variance.numpy())
        return float(self.scale.numpy()) ** 2

    def variance_at(self, X) -> torch.Tensor:
        X = ops.to_device(X)
        return torch.full(X.shape[:-1] + (1,), self.noise_variance(), dtype=torch.float64, device=X.device)

    def log_prob(self, X, F, Y):
        return gaussian(ops.to_device(Y), ops.to_device(F), self.noise_variance()).sum(-1)

    def predict_mean_and_var(self, X, Fmu, Fvar):
        """scalar_continuous.py:127-130"""
        return Fmu.clone(), Fvar + self.noise_variance()

    def predict_log_density(self, X, Fmu, Fvar, Y):
        """scalar_continuous.py:132-136"""
        return gaussian(ops.to_device(Y), Fmu, Fvar + self.noise_variance()).sum(-1)

    def variational_expectations(self, X, Fmu, Fvar, Y) -> torch.Tensor:
        """scalar_continuous.py:139-148 -- per-row values [N] (elementwise glue; the summed form used
        by SVGP.elbo runs in gpk_gaussian_varexp_sum)."""
        v = self.noise_variance()
        Y = ops.to_device(Y)
        return (-0.5 * LOG2PI - 0.5 * float(np.log(v)) - 0.5 * ((Y - Fmu) ** 2 + Fvar) / v).sum(-1)
