"""Kernel ABC: the __call__ contract and active_dims slicing of gpflow/kernels/base.py:90-214."""
from __future__ import annotations

import abc
from typing import Optional, Sequence, Union

import numpy as np
import torch

from ..base import Module
from .. import ops

ActiveDims = Union[slice, Sequence[int]]


class Kernel(Module, metaclass=abc.ABCMeta):
    def __init__(self, active_dims: Optional[ActiveDims] = None, name: Optional[str] = None):
        self.name = name or type(self).__name__
        if active_dims is None:
            active_dims = slice(None, None, None)
        if not isinstance(active_dims, slice):
            active_dims = np.array(active_dims, dtype=int)
        self._active_dims = active_dims

    @property
    def active_dims(self):
        return self._active_dims

    def slice(self, X: torch.Tensor, X2: Optional[torch.Tensor] = None):
        """gpflow/kernels/base.py:90-109"""
        dims = self._active_dims
        if isinstance(dims, slice):
            if dims == slice(None, None, None):
                return X, X2
            X = X[..., dims]
            X2 = X2[..., dims] if X2 is not None else None
        else:
            idx = torch.as_tensor(dims, device=X.device)
            X = X.index_select(-1, idx)
            X2 = X2.index_select(-1, idx) if X2 is not None else None
        return X.contiguous(), (X2.contiguous() if X2 is not None else None)

    @abc.abstractmethod
    def K(self, X, X2=None):
        raise NotImplementedError

    @abc.abstractmethod
    def K_diag(self, X):
        raise NotImplementedError

    def __call__(self, X, X2=None, *, full_cov: bool = True, presliced: bool = False):
        """gpflow/kernels/base.py:195-214"""
        if (not full_cov) and (X2 is not None):
            raise ValueError("Ambiguous inputs: `not full_cov` and `X2` are not compatible.")
        X = ops.to_device(X)
        X2 = ops.to_device(X2) if X2 is not None else None
        if not presliced:
            X, X2 = self.slice(X, X2)
        if not full_cov:
            return self.K_diag(X)
        return self.K(X, X2)
