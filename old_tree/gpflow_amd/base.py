"""TF-free Parameter / Module substrate (surface of gpflow/base.py:73-330).

A Parameter keeps the UNCONSTRAINED value on the host (NumPy fp64) plus a bijector, exactly like
tfp.util.TransformedVariable; the constrained value is what the device kernels see.  Array-valued
parameters (Z, q_mu, q_sqrt) additionally keep a lazily refreshed device copy.
"""
from __future__ import annotations

from enum import Enum
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np

from . import config

# ------------------------------------------------------------------------------------------ bijectors


class Bijector:
    name = "bijector"

    def forward(self, x: np.ndarray) -> np.ndarray:  # unconstrained -> constrained
        raise NotImplementedError

    def inverse(self, y: np.ndarray) -> np.ndarray:
        raise NotImplementedError

    def forward_grad(self, x: np.ndarray) -> np.ndarray:
        """Elementwise d forward(x) / dx (the chain-rule factor from constrained to unconstrained gradients; TF
        autodiff supplies it in the reference, base.py:137-280 + tfp bijectors)."""
        raise NotImplementedError(f"{self.name} has no elementwise derivative")


class Identity(Bijector):
    name = "identity"

    def forward(self, x):
        return np.asarray(x, dtype=np.float64)

    def forward_grad(self, x):
        return np.ones_like(np.asarray(x, dtype=np.float64))

    def inverse(self, y):
        return np.asarray(y, dtype=np.float64)


class Softplus(Bijector):
    name = "softplus"

    def forward(self, x):
        return np.logaddexp(0.0, np.asarray(x, dtype=np.float64))

    def forward_grad(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.exp(-np.logaddexp(0.0, -x))  # sigmoid(x)

    def inverse(self, y):
        y = np.asarray(y, dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            return y + np.log(-np.expm1(-y))  # tfp.math.softplus_inverse


class Exp(Bijector):
    name = "exp"

    def forward(self, x):
        return np.exp(np.asarray(x, dtype=np.float64))

    def forward_grad(self, x):
        return np.exp(np.asarray(x, dtype=np.float64))

    def inverse(self, y):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.log(np.asarray(y, dtype=np.float64))


class Shift(Bijector):
    name = "shift"

    def __init__(self, shift: float):
        self.shift = float(shift)

    def forward(self, x):
        return np.asarray(x, dtype=np.float64) + self.shift

    def forward_grad(self, x):
        return np.ones_like(np.asarray(x, dtype=np.float64))

    def inverse(self, y):
        return np.asarray(y, dtype=np.float64) - self.shift


class Chain(Bijector):
    """Chain([b1, b2]).forward(x) = b1.forward(b2.forward(x)) (tfp.bijectors.Chain order)."""

    def __init__(self, bijectors: List[Bijector]):
        self.bijectors = list(bijectors)
        self.name = "chain_of_" + "_of_".join(b.name for b in self.bijectors)

    def forward(self, x):
        for b in reversed(self.bijectors):
            x = b.forward(x)
        return x

    def forward_grad(self, x):
        g = np.ones_like(np.asarray(x, dtype=np.float64))
        for b in reversed(self.bijectors):
            g = g * b.forward_grad(x)
            x = b.forward(x)
        return g

    def inverse(self, y):
        for b in self.bijectors:
            y = b.inverse(y)
        return y


class FillTriangular(Bijector):
    """tfp.bijectors.FillTriangular: vector [..., n(n+1)/2] <-> lower-triangular [..., n, n]."""

    name = "fill_triangular"

    @staticmethod
    def _grid(n: int) -> np.ndarray:
        m = n * (n + 1) // 2
        idx = np.arange(m)
        return np.concatenate([idx[n:], idx[::-1]]).reshape(n, n)

    def forward(self, x):
        x = np.asarray(x, dtype=np.float64)
        m = x.shape[-1]
        n = int(round(np.sqrt(0.25 + 2.0 * m) - 0.5))
        if n * (n + 1) // 2 != m:
            raise ValueError(f"vector length {m} is not a triangular number")
        xc = np.concatenate([x[..., n:], x[..., ::-1]], axis=-1)
        return np.tril(xc.reshape(x.shape[:-1] + (n, n)))

    def inverse(self, y):
        y = np.asarray(y, dtype=np.float64)
        n = y.shape[-1]
        r, c = np.tril_indices(n)
        out = np.zeros(y.shape[:-2] + (n * (n + 1) // 2,), dtype=np.float64)
        out[..., self._grid(n)[r, c]] = y[..., r, c]
        return out


def positive(lower: Optional[float] = None, base: Optional[str] = None) -> Bijector:
    """gpflow/utilities/bijectors.py:27-45"""
    name = (base if base is not None else config.default_positive_bijector()).lower()
    if name not in ("softplus", "exp"):
        raise KeyError(name)
    bij: Bijector = Softplus() if name == "softplus" else Exp()
    lower_bound = lower if lower is not None else config.default_positive_minimum()
    if lower_bound != 0.0:
        bij = Chain([Shift(lower_bound), bij])
    return bij


def triangular() -> Bijector:
    """gpflow/utilities/bijectors.py:48-52"""
    return FillTriangular()


# ------------------------------------------------------------------------------------------ Parameter


class PriorOn(Enum):
    CONSTRAINED = "constrained"
    UNCONSTRAINED = "unconstrained"


def _to_numpy(value: Any) -> np.ndarray:
    if isinstance(value, Parameter):
        return value.numpy()
    try:
        import torch

        if isinstance(value, torch.Tensor):
            return value.detach().cpu().numpy().astype(np.float64)
    except ImportError:  # pragma: no cover
        pass
    return np.asarray(value, dtype=np.float64)


def _validate_unconstrained(value: Any, transform: Bijector) -> np.ndarray:
    """gpflow/base.py:314-326: the unconstrained value must be finite."""
    unconstrained = np.asarray(transform.inverse(_to_numpy(value)), dtype=np.float64)
    if not np.all(np.isfinite(unconstrained)):
        raise ValueError(
            "gpflow.Parameter: the value to be assigned is incompatible with this parameter's "
            "transform (the corresponding unconstrained value has NaN or Inf) and hence cannot be "
            "assigned."
        )
    return unconstrained


class Parameter:
    """gpflow/base.py:118-280 without TensorFlow: value, transform, prior, trainable."""

    def __init__(self, value: Any, *, transform: Optional[Bijector] = None, prior: Any = None,
                 prior_on: Any = None, trainable: Optional[bool] = None, dtype: Any = None,
                 name: Optional[str] = None, unconstrained_shape=None, constrained_shape=None,
                 shape=None):
        if isinstance(value, Parameter):
            transform = transform or value.transform
            prior = prior or value.prior
            prior_on = prior_on or value.prior_on
            trainable = value.trainable if trainable is None else trainable
            name = name or value.name
        if transform is None:
            transform = Identity()
        if dtype is not None and np.dtype(dtype) != np.float64:
            raise TypeError("gpflow_amd parameters are float64")
        self._transform = transform
        self._unconstrained = _validate_unconstrained(value, transform)
        self.prior = prior
        self.prior_on = PriorOn(prior_on) if prior_on else PriorOn.CONSTRAINED
        self._trainable = True if trainable is None else bool(trainable)
        self.name = name or transform.name
        self._device_cache = None

    # -- values --------------------------------------------------------------------------------
    def numpy(self) -> np.ndarray:
        return np.asarray(self._transform.forward(self._unconstrained), dtype=np.float64)

    def __array__(self, dtype=None, copy=None):
        v = self.numpy()
        return v.astype(dtype) if dtype is not None else v

    def __float__(self) -> float:
        return float(self.numpy())

    @property
    def shape(self) -> Tuple[int, ...]:
        return self.numpy().shape

    @property
    def dtype(self):
        return np.float64

    @property
    def transform(self) -> Bijector:
        return self._transform

    bijector = transform

    @property
    def unconstrained_variable(self) -> np.ndarray:
        return self._unconstrained

    @property
    def trainable(self) -> bool:
        return self._trainable

    def assign(self, value: Any) -> "Parameter":
        """Assign a CONSTRAINED value (gpflow/base.py:253-280)."""
        new = _validate_unconstrained(value, self._transform)
        if new.shape != self._unconstrained.shape:
            raise ValueError(f"shape mismatch: parameter {self._unconstrained.shape}, value {new.shape}")
        self._unconstrained = new
        self._device_cache = None
        return self

    def assign_unconstrained(self, value: Any) -> "Parameter":
        new = np.asarray(value, dtype=np.float64).reshape(self._unconstrained.shape)
        self._unconstrained = new
        self._device_cache = None
        return self

    def device_value(self):
        """Constrained value as a contiguous fp64 tensor on the HIP device (cached until assign)."""
        if self._device_cache is None:
            from . import ops

            self._device_cache = ops.to_device(self.numpy())
        return self._device_cache

    def log_prior_density(self) -> float:
        """gpflow/base.py:201-224: log density of the prior, evaluated on the constrained value, or -- `prior_on=UNCONSTRAINED`
        -- on the unconstrained one plus the log|Jacobian| of the inverse transform (the density is reported in the
        constrained space either way)."""
        if self.prior is None:
            return 0.0
        if self.prior_on == PriorOn.CONSTRAINED:
            return float(np.sum(self.prior.log_prob(self.numpy())))
        x = self._unconstrained
        # inverse_log_det_jacobian(y) = log |dx/dy| = -log |d forward(x)/dx|, elementwise transforms
        return float(np.sum(self.prior.log_prob(x))) - float(np.sum(np.log(np.abs(self._transform.forward_grad(x)))))

    def log_prior_density_grad(self) -> np.ndarray:
        """d log_prior_density / d(unconstrained value): what TF autodiff adds to every training-loss gradient in the
        reference (models/model.py:47-76 -> optimizers/scipy.py:322-331).  Shape of the unconstrained value."""
        from .priors import grad_log_prob
        x = self._unconstrained
        if self.prior is None:
            return np.zeros_like(x)
        fg = self._transform.forward_grad(x)   # raises for non-elementwise transforms (FillTriangular)
        if self.prior_on == PriorOn.CONSTRAINED:
            return grad_log_prob(self.prior, self._transform.forward(x)) * fg
        # unconstrained prior: d/dx [ log p(x) - log |fg(x)| ]; the second term by central differences of the transform
        h = 1e-5 * np.maximum(1.0, np.abs(x))
        dlogfg = (np.log(np.abs(self._transform.forward_grad(x + h))) - np.log(np.abs(self._transform.forward_grad(x - h)))) / (2.0 * h)
        return grad_log_prob(self.prior, x) - dlogfg

    def __repr__(self) -> str:
        return f"<Parameter name={self.name} shape={self.shape} value={self.numpy()!r}>"


def set_trainable(model: Any, flag: bool) -> None:
    """gpflow/utilities/misc.py:set_trainable"""
    params = [model] if isinstance(model, Parameter) else list(model.parameters)
    for p in params:
        p._trainable = bool(flag)


class Module:
    """gpflow/base.py:73-110: attribute traversal for parameters."""

    def _walk(self, prefix: str = "", seen=None) -> Iterator[Tuple[str, Parameter]]:
        seen = set() if seen is None else seen
        if id(self) in seen:
            return
        seen.add(id(self))
        for key, val in vars(self).items():
            if key.startswith("_Module__"):
                continue
            path = f"{prefix}.{key}"
            yield from _walk_value(val, path, seen)

    @property
    def parameters(self) -> Tuple[Parameter, ...]:
        out, ids = [], set()
        for _, p in self._walk():
            if id(p) not in ids:
                ids.add(id(p))
                out.append(p)
        return tuple(out)

    @property
    def trainable_parameters(self) -> Tuple[Parameter, ...]:
        return tuple(p for p in self.parameters if p.trainable)


def _walk_value(val: Any, path: str, seen) -> Iterator[Tuple[str, Parameter]]:
    if isinstance(val, Parameter):
        yield path, val
    elif isinstance(val, Module):
        yield from val._walk(path, seen)
    elif isinstance(val, (list, tuple)):
        for i, v in enumerate(val):
            yield from _walk_value(v, f"{path}[{i}]", seen)
    elif isinstance(val, dict):
        for k, v in val.items():
            yield from _walk_value(v, f"{path}['{k}']", seen)


def parameter_dict(module: Module) -> Dict[str, Parameter]:
    """gpflow/utilities/traversal.py:52-64"""
    return {path: p for path, p in module._walk("")}


def read_values(module: Module) -> Dict[str, np.ndarray]:
    """gpflow/utilities/traversal.py:85-92"""
    return {k: p.numpy() for k, p in parameter_dict(module).items()}


def multiple_assign(module: Module, values: Dict[str, Any]) -> None:
    """gpflow/utilities/traversal.py:67-82"""
    refs = parameter_dict(module)
    for k, v in values.items():
        refs[k].assign(v)
