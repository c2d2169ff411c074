"""Mean functions on the path (gpflow/functions.py:173-204): Zero and Constant."""
from __future__ import annotations

import numpy as np
import torch

from .base import Module, Parameter
from . import ops


class MeanFunction(Module):
    def __call__(self, X):
        raise NotImplementedError

    def constant_value(self):
        """Scalar c such that m(X) == c everywhere, or None (enables the fused device paths)."""
        return None


class Zero(MeanFunction):
    """functions.py:195-204"""

    def __init__(self, output_dim: int = 1):
        self.output_dim = output_dim

    def __call__(self, X):
        X = ops.to_device(X)
        return torch.zeros(X.shape[:-1] + (self.output_dim,), dtype=torch.float64, device=X.device)

    def constant_value(self):
        return 0.0


class Constant(MeanFunction):
    """functions.py:173-192"""

    def __init__(self, c=None):
        c = np.zeros(1) if c is None else c
        self.c = Parameter(c)

    def __call__(self, X):
        X = ops.to_device(X)
        c = ops.to_device(np.atleast_1d(self.c.numpy()))
        return torch.ones(X.shape[:-1] + (1,), dtype=torch.float64, device=X.device) * c

    def constant_value(self):
        c = np.atleast_1d(self.c.numpy())
        return float(c[0]) if c.size == 1 else None
