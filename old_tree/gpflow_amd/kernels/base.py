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

    @property
    def has_default_active_dims(self) -> bool:
        """True when every input column is active (active_dims was None / slice(None)); an index list is never the
        default, even if it happens to enumerate all columns."""
        d = self._active_dims
        return isinstance(d, slice) and d == slice(None, None, None)

    def slice(self, X: torch.Tensor, X2: Optional[torch.Tensor] = None):
        """gpflow/kernels/base.py:90-109"""
        dims = self._active_dims
        if isinstance(dims, slice):
            if self.has_default_active_dims:
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

    def K_into(self, X, X2, out, *, diag_add: float = 0.0, lower_only: bool = False):
        """Device contract used by the covariance dispatchers, conditionals and posteriors: K(X, X2) (K(X, X) +
        diag_add I when X2 is None) written into `out` (allocated when None); inputs are already sliced by the caller.
        Stationary kernels and their sums / products override this with fused device builds; this generic version
        only serves kernels that define K alone."""
        K = self.K(X, X2)
        if X2 is None and diag_add != 0.0:
            K.diagonal().add_(float(diag_add))
        if out is None:
            return K
        out.copy_(K)
        return out

    def __add__(self, other: "Kernel") -> "Kernel":
        """gpflow/kernels/base.py:216-217"""
        return Sum([self, other])

    def __mul__(self, other: "Kernel") -> "Kernel":
        """gpflow/kernels/base.py:219-220"""
        return Product([self, other])

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


class Combination(Kernel):
    """A list of kernels reduced elementwise (gpflow/kernels/base.py:223-329); nested instances of the same class are
    flattened (:247-255).  Every member slices its own active_dims, so the combination itself never slices."""

    _op = None  # "add" | "mul"

    def __init__(self, kernels: Sequence[Kernel], name: Optional[str] = None):
        super().__init__(name=name)
        if not all(isinstance(k, Kernel) for k in kernels):
            raise TypeError("can only combine Kernel instances")
        flat = []
        for k in kernels:
            flat.extend(k.kernels if isinstance(k, self.__class__) else [k])
        self.kernels = flat

    @property
    def on_separate_dimensions(self) -> bool:
        """gpflow/kernels/base.py:257-280"""
        if any(isinstance(k.active_dims, slice) for k in self.kernels):
            return False
        dims = [np.asarray(k.active_dims) for k in self.kernels]
        for i, di in enumerate(dims):
            for dj in dims[i + 1:]:
                if np.any(di.reshape(-1, 1) == dj.reshape(1, -1)):
                    return False
        return True

    def K_into(self, X, X2, out, *, diag_add: float = 0.0, lower_only: bool = False):
        """The first member is built into `out`; every further stationary member is folded in by
        gpk_kernel_matrix_combine (K recomputed in registers: one read + one write of `out`, no second matrix)."""
        from .stationaries import Stationary
        X = ops.to_device(X)
        X2 = ops.to_device(X2) if X2 is not None else None
        ks = list(self.kernels)
        last = len(ks) - 1
        k0 = ks[0]
        Xs, X2s = k0.slice(X, X2)
        out = k0.K_into(Xs, X2s, out, diag_add=diag_add if last == 0 else 0.0, lower_only=False)
        for i, k in enumerate(ks[1:], start=1):
            Xs, X2s = k.slice(X, X2)
            dadd = diag_add if i == last else 0.0
            if isinstance(k, Stationary):
                family, var, ls = k.hyper()
                ops.kernel_matrix_combine(Xs, X2s, out, op=self._op, variance=var, lengthscales=ls, family=family,
                                          diag_add=dadd, out=out)
            else:
                Ki = k.K_into(Xs, X2s, None)
                out.add_(Ki) if self._op == "add" else out.mul_(Ki)
                if X2 is None and dadd != 0.0:
                    out.diagonal().add_(float(dadd))
        return out

    def K(self, X, X2=None) -> torch.Tensor:
        X = ops.to_device(X)
        if X.dim() != 2 or (X2 is not None and ops.to_device(X2).dim() != 2):
            raise NotImplementedError("kernel combinations take [N, D] inputs")
        return self.K_into(X, X2, None)

    def K_diag(self, X) -> torch.Tensor:
        outs = [k(X, full_cov=False) for k in self.kernels]
        acc = outs[0].clone()
        for o in outs[1:]:
            acc = acc + o if self._op == "add" else acc * o
        return acc

    def __call__(self, X, X2=None, *, full_cov: bool = True, presliced: bool = False):
        """gpflow/kernels/base.py:283-293: members slice for themselves unless presliced."""
        if (not full_cov) and (X2 is not None):
            raise ValueError("Ambiguous inputs: `not full_cov` and `X2` are not compatible.")
        if presliced:
            raise NotImplementedError("presliced inputs to a kernel combination")
        return self.K_diag(X) if not full_cov else self.K(X, X2)


def gradient_spec(kernel, input_dim=None):
    """gradients.KernelSpec + [(variance Parameter, lengthscales Parameter)] for the kernels the reverse pass covers beyond a
    single stationary one: a Sum / Product of isotropic-stationary members, flat or nested (kernels/base.py:216-220, 305-315; the
    end of round 5: a Product of Sums and the like -- the spec then carries the combination tree).  Members may
    carry their own `active_dims` (kernels/base.py:90-109): the spec then builds and differentiates each member on its own columns
    and scatters its input gradient back (round 5) -- `input_dim`, the number of input columns, resolves slices.  None if `kernel`
    is not such a combination."""
    from .stationaries import IsotropicStationary
    from .. import gradients
    if not isinstance(kernel, Combination):
        return None
    # leaves in traversal order + the tree over their indices (a flat combination: one node)
    ks = []

    def walk(k):
        if isinstance(k, Combination):
            return (k._op, [walk(c) for c in k.kernels])
        if not (isinstance(k, IsotropicStationary) and k.family in ops.KERNEL_FAMILIES):
            raise NotImplementedError("gradients of a kernel combination: Sums / Products (possibly nested) of SquaredExponential / "
                                      "Matern members (kernels/base.py:216-220, 305-315)")
        ks.append(k)
        return len(ks) - 1
    tree = walk(kernel)
    nested = any(not isinstance(c, int) for c in tree[1])
    cols = []
    for k in ks:
        if k.has_default_active_dims:
            cols.append(None)
        elif isinstance(k.active_dims, slice):
            if input_dim is None:
                raise NotImplementedError("gradients of a kernel combination with sliced active_dims need the input dimension")
            cols.append(np.arange(int(input_dim))[k.active_dims])
        else:
            cols.append(np.asarray(k.active_dims, dtype=np.int64))
    spec = gradients.KernelSpec([k.hyper() for k in ks], tree if nested else kernel._op, cols=cols)
    return spec, [(k.variance, k.lengthscales) for k in ks]


class Sum(Combination):
    """gpflow/kernels/base.py:318-321"""
    _op = "add"


class Product(Combination):
    """gpflow/kernels/base.py:324-329"""
    _op = "mul"
