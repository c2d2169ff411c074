"""Functions of the inputs (gpflow/functions.py:38-330): mean functions AND the variance / scale of a heteroskedastic Gaussian
likelihood (likelihoods/scalar_continuous.py:52-111).  Each row of X is one datum; f(X) has one row per datum.  O(N D Q) work --
nothing here is on the O(N^3) / O(M^2 B) path.  The matrix products (X A, the polynomial's feature-weight product and their
reverse passes) go through the library's own GEMM (gpk_gemm_nt) like every other product of the package: no rocBLAS call is made
from this file; what is left to torch is elementwise (powers, sums, broadcasts).  `constant_value()` tells the fused C-ABI drivers
when a function is a scalar constant so that it can ride inside them."""
from __future__ import annotations

import itertools
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .base import Module, Parameter


def _mm_nt(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A [n, k] times B[q, k]^T -> [n, q] on the library's GEMM (rows of both operands contiguous)."""
    if A.shape[0] == 0 or B.shape[0] == 0:
        return torch.zeros((A.shape[0], B.shape[0]), dtype=torch.float64, device=A.device)
    return ops.gemm_nt(A.contiguous(), B.contiguous())


class Function(Module):
    """functions.py:38-64"""

    def __call__(self, X):
        raise NotImplementedError("Implement the __call__ method for this mean function")

    def __add__(self, other: "Function") -> "Function":
        return Additive(self, other)

    def __mul__(self, other: "Function") -> "Function":
        return Product(self, other)

    def constant_value(self):
        """Scalar c such that f(X) == c everywhere, or None (enables the fused device paths)."""
        return None

    def backward(self, X, gbar):
        """[(Parameter, dF/d(constrained value) as a device tensor)] given gbar = dF/d f(X) [N, Q]: the reverse pass of the function
        (TF autodiff in the reference), written out for the classes here -- used for the parameters of a heteroskedastic noise
        function (likelihoods.Gaussian.noise_param_grads)."""
        raise NotImplementedError(f"reverse pass of {type(self).__name__}")


class MeanFunction(Function):
    """functions.py:67-70"""


class Additive(MeanFunction):
    """functions.py:73-81"""

    def __init__(self, first_part: Function, second_part: Function):
        self.add_1 = first_part
        self.add_2 = second_part

    def __call__(self, X):
        return self.add_1(X) + self.add_2(X)

    def backward(self, X, gbar):
        return self.add_1.backward(X, gbar) + self.add_2.backward(X, gbar)

    def constant_value(self):
        a, b = self.add_1.constant_value(), self.add_2.constant_value()
        return None if a is None or b is None else a + b


class Product(MeanFunction):
    """functions.py:84-93"""

    def __init__(self, first_part: Function, second_part: Function):
        self.prod_1 = first_part
        self.prod_2 = second_part

    def __call__(self, X):
        return self.prod_1(X) * self.prod_2(X)

    def backward(self, X, gbar):
        f1, f2 = self.prod_1(X), self.prod_2(X)
        q = gbar.shape[-1]
        red = lambda g, f: g if f.shape[-1] == q else g.sum(-1, keepdim=True)  # noqa: E731  (a [N, 1] factor broadcast over Q)
        return self.prod_1.backward(X, red(gbar * f2, f1)) + self.prod_2.backward(X, red(gbar * f1, f2))

    def constant_value(self):
        a, b = self.prod_1.constant_value(), self.prod_2.constant_value()
        return None if a is None or b is None else a * b


class Linear(MeanFunction):
    """y_i = A x_i + b (functions.py:96-126): A [D, Q] (default ones [1, 1]), b [Q] (default zeros [1])."""

    def __init__(self, A=None, b=None):
        A = np.ones((1, 1)) if A is None else A
        b = np.zeros(1) if b is None else b
        if isinstance(A, Parameter):
            if len(A.shape) < 2:
                raise ValueError("Error 'gpflow.funcitons.Linear()' mean function. A has not the correct shape (at least 2d).")
            self.A = A
        else:
            self.A = Parameter(np.atleast_2d(np.asarray(A, dtype=np.float64)))
        self.b = b if isinstance(b, Parameter) else Parameter(b)

    def __call__(self, X):
        X = ops.to_device(X)
        A = ops.to_device(np.asarray(self.A.numpy(), dtype=np.float64))
        b = ops.to_device(np.atleast_1d(np.asarray(self.b.numpy(), dtype=np.float64)))
        lead = X.shape[:-1]
        return _mm_nt(X.reshape(-1, X.shape[-1]), A.t()).reshape(lead + (A.shape[1],)) + b     # X A (+ b), leading dims flattened

    def backward(self, X, gbar):
        X = ops.to_device(X)
        gA = _mm_nt(X.t(), gbar.t())                            # X^T gbar  [D, Q']: Q' = Q, or 1 when A broadcasts over the outputs
        A_shape, b_shape = tuple(self.A.shape), tuple(np.atleast_1d(self.b.numpy()).shape)
        if gA.shape[1] != A_shape[1]:
            gA = gA.sum(1, keepdim=True)
        if gA.shape[0] != A_shape[0]:
            gA = gA.sum(0, keepdim=True)
        gb = gbar.sum(0)
        if gb.numel() != int(np.prod(b_shape)):
            gb = gb.sum().reshape(1)
        return [(self.A, gA), (self.b, gb.reshape(self.b.numpy().shape))]


class Identity(Linear):
    """y_i = x_i (functions.py:129-170)"""

    def __init__(self, input_dim: Optional[int] = None):
        self.input_dim = input_dim

    def __call__(self, X):
        return ops.to_device(X)

    def backward(self, X, gbar):
        return []

    def _need_dim(self):
        if self.input_dim is None:
            raise ValueError("An input_dim needs to be specified when using the `Identity` mean function in combination "
                             "with expectations.")

    @property
    def A(self):
        self._need_dim()
        return torch.eye(self.input_dim, dtype=torch.float64)

    @A.setter
    def A(self, value):
        pass

    @property
    def b(self):
        self._need_dim()
        return torch.zeros(self.input_dim, dtype=torch.float64)

    @b.setter
    def b(self, value):
        pass


class Constant(MeanFunction):
    """functions.py:173-192"""

    def __init__(self, c=None):
        c = np.zeros(1) if c is None else c
        self.c = c if isinstance(c, Parameter) else Parameter(c)

    def __call__(self, X):
        X = ops.to_device(X)
        c = ops.to_device(np.atleast_1d(self.c.numpy()))
        return torch.ones(X.shape[:-1] + (1,), dtype=torch.float64, device=X.device) * c

    def constant_value(self):
        c = np.atleast_1d(self.c.numpy())
        return float(c[0]) if c.size == 1 else None

    def backward(self, X, gbar):
        g = gbar.sum(0)
        n = int(np.size(self.c.numpy()))
        return [(self.c, (g if g.numel() == n else g.sum().reshape(1)).reshape(self.c.numpy().shape))]


class Zero(Constant):
    """functions.py:195-204"""

    def __init__(self, output_dim: int = 1):
        self.output_dim = output_dim

    def __call__(self, X):
        X = ops.to_device(X)
        return torch.zeros(X.shape[:-1] + (self.output_dim,), dtype=torch.float64, device=X.device)

    def constant_value(self):
        return 0.0

    def backward(self, X, gbar):
        return []


class Polynomial(MeanFunction):
    """A generic polynomial (functions.py:207-278): f(x)_j = sum_i w[j, i] prod_d x_d ** powers[i, d]."""

    def __init__(self, degree: int, input_dim: int = 1, output_dim: int = 1, w=None):
        powers = self.compute_powers(degree, input_dim)
        if w is None:
            w = [1.0] + (len(powers) - 1) * [0.0]
        self.powers = np.asarray(powers, dtype=np.float64).reshape(len(powers), input_dim)
        self.w = Parameter(np.broadcast_to(np.asarray(w, dtype=np.float64), (output_dim, len(powers))).copy())

    @staticmethod
    def compute_powers(degree: int, input_dim: int) -> Sequence[Tuple[int, ...]]:
        """All exponent tuples of length input_dim with non-negative entries summing to at most `degree`, in lexicographical
        order (functions.py:229-272)."""
        return [t for t in itertools.product(range(degree + 1), repeat=input_dim) if sum(t) <= degree]

    def __call__(self, X):
        X = ops.to_device(X)
        powers = ops.to_device(self.powers)
        raised = torch.pow(X[..., None, :], powers)            # [..., n_terms, input_dim]
        prod = torch.prod(raised, dim=-1)                      # [..., n_terms]
        w = ops.to_device(np.asarray(self.w.numpy(), dtype=np.float64))                        # [Q, n_terms]
        lead = prod.shape[:-1]
        return _mm_nt(prod.reshape(-1, prod.shape[-1]), w).reshape(lead + (w.shape[0],))       # sum_i prod[..., i] w[j, i]

    def backward(self, X, gbar):
        X = ops.to_device(X)
        prod = torch.prod(torch.pow(X[..., None, :], ops.to_device(self.powers)), dim=-1)      # [N, n_terms]
        gw = _mm_nt(gbar.t(), prod.t())                                                         # gbar^T prod  [Q', n_terms]
        if gw.shape[0] != self.w.shape[0]:
            gw = gw.expand(self.w.shape[0], -1) if gw.shape[0] == 1 else gw.sum(0, keepdim=True)
        return [(self.w, gw)]
