"""Parameter priors for MAP estimation (gpflow/base.py:197-224, models/model.py:47-76): the reference hangs a
`tfp.distributions.Distribution` on a Parameter and adds its log density to every training loss; TF autodiff then
differentiates it.  Here a prior is any object with `log_prob(x) -> ndarray` (elementwise log density); if it also has
`grad_log_prob(x)` the reverse pass uses that, otherwise it differentiates `log_prob` by central differences on the host
(priors act on a handful of scalars).  The closed forms below cover the distributions GPflow's own notebooks use.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.special import gammaln


def _a(x):
    return np.asarray(x, dtype=np.float64)


def _t(v, like):
    """a prior's (host) parameter as a tensor beside `like`"""
    return torch.as_tensor(np.asarray(v, dtype=np.float64), dtype=like.dtype, device=like.device)


class Normal:
    """tfp.distributions.Normal(loc, scale)"""

    def __init__(self, loc=0.0, scale=1.0):
        self.loc, self.scale = _a(loc), _a(scale)

    def log_prob(self, x):
        z = (_a(x) - self.loc) / self.scale
        return -0.5 * z * z - np.log(self.scale) - 0.5 * np.log(2.0 * np.pi)

    def grad_log_prob(self, x):
        return -(_a(x) - self.loc) / (self.scale * self.scale)

    def log_prob_torch(self, x):
        loc, scale = _t(self.loc, x), _t(self.scale, x)
        z = (x - loc) / scale
        return -0.5 * z * z - torch.log(scale) - 0.5 * float(np.log(2.0 * np.pi))

    def grad_log_prob_torch(self, x):
        loc, scale = _t(self.loc, x), _t(self.scale, x)
        return -(x - loc) / (scale * scale)


class Gamma:
    """tfp.distributions.Gamma(concentration, rate): density x^(a-1) exp(-b x) b^a / Gamma(a)"""

    def __init__(self, concentration, rate):
        self.concentration, self.rate = _a(concentration), _a(rate)

    def log_prob(self, x):
        x = _a(x)
        a, b = self.concentration, self.rate
        return (a - 1.0) * np.log(x) - b * x + a * np.log(b) - gammaln(a)

    def grad_log_prob(self, x):
        return (self.concentration - 1.0) / _a(x) - self.rate

    def log_prob_torch(self, x):
        a, b = _t(self.concentration, x), _t(self.rate, x)
        return (a - 1.0) * torch.log(x) - b * x + a * torch.log(b) - torch.lgamma(a)

    def grad_log_prob_torch(self, x):
        return (_t(self.concentration, x) - 1.0) / x - _t(self.rate, x)


class LogNormal:
    """tfp.distributions.LogNormal(loc, scale)"""

    def __init__(self, loc=0.0, scale=1.0):
        self.loc, self.scale = _a(loc), _a(scale)

    def log_prob(self, x):
        x = _a(x)
        z = (np.log(x) - self.loc) / self.scale
        return -0.5 * z * z - np.log(self.scale * x) - 0.5 * np.log(2.0 * np.pi)

    def grad_log_prob(self, x):
        x = _a(x)
        return -(np.log(x) - self.loc) / (self.scale * self.scale * x) - 1.0 / x

    def log_prob_torch(self, x):
        loc, scale = _t(self.loc, x), _t(self.scale, x)
        z = (torch.log(x) - loc) / scale
        return -0.5 * z * z - torch.log(scale * x) - 0.5 * float(np.log(2.0 * np.pi))

    def grad_log_prob_torch(self, x):
        loc, scale = _t(self.loc, x), _t(self.scale, x)
        return -(torch.log(x) - loc) / (scale * scale * x) - 1.0 / x


class HalfNormal:
    """tfp.distributions.HalfNormal(scale)"""

    def __init__(self, scale=1.0):
        self.scale = _a(scale)

    def log_prob(self, x):
        x = _a(x)
        lp = -0.5 * (x / self.scale) ** 2 + 0.5 * np.log(2.0 / np.pi) - np.log(self.scale)
        return np.where(x >= 0.0, lp, -np.inf)

    def grad_log_prob(self, x):
        return -_a(x) / (self.scale * self.scale)

    def log_prob_torch(self, x):
        scale = _t(self.scale, x)
        lp = -0.5 * (x / scale) ** 2 + 0.5 * float(np.log(2.0 / np.pi)) - torch.log(scale)
        return torch.where(x >= 0.0, lp, torch.full_like(lp, -float("inf")))

    def grad_log_prob_torch(self, x):
        scale = _t(self.scale, x)
        return -x / (scale * scale)


def grad_log_prob(prior, x) -> np.ndarray:
    """d sum(prior.log_prob(x)) / dx, elementwise: the prior's own `grad_log_prob` if it has one, else Richardson-extrapolated
    central differences of `log_prob` (one element at a time, so priors that couple elements are handled too)."""
    x = np.array(x, dtype=np.float64)
    if hasattr(prior, "grad_log_prob"):
        return np.asarray(prior.grad_log_prob(x), dtype=np.float64).reshape(x.shape)
    g = np.zeros_like(x)
    f = lambda v: float(np.sum(prior.log_prob(v)))  # noqa: E731
    for idx in np.ndindex(*x.shape) if x.shape else [()]:
        h = 1e-4 * max(1.0, abs(float(x[idx])))
        d = []
        for hh in (h, 0.5 * h):
            xp, xm = x.copy(), x.copy()
            xp[idx] += hh
            xm[idx] -= hh
            d.append((f(xp) - f(xm)) / (2.0 * hh))
        g[idx] = (4.0 * d[1] - d[0]) / 3.0
    return g


# ---- priors on DEVICE-resident variables (training.SVGPTrainer keeps Z, q_mu, q_sqrt and their Adam moments in HBM) --------------
def log_prob_device(prior, x: "torch.Tensor") -> "torch.Tensor":
    """sum(prior.log_prob(x)) as a 0-d tensor on x's device: the closed forms above run there (elementwise torch glue); any other
    prior object goes through the host once per call (it is user code over NumPy)."""
    if hasattr(prior, "log_prob_torch"):
        return prior.log_prob_torch(x).sum()
    return torch.as_tensor(float(np.sum(prior.log_prob(x.detach().cpu().numpy()))), dtype=x.dtype, device=x.device)


def grad_log_prob_device(prior, x: "torch.Tensor") -> "torch.Tensor":
    """d sum(prior.log_prob(x)) / dx, shape of x, on x's device (see log_prob_device)."""
    if hasattr(prior, "grad_log_prob_torch"):
        return prior.grad_log_prob_torch(x)
    return torch.as_tensor(grad_log_prob(prior, x.detach().cpu().numpy()), dtype=x.dtype, device=x.device)
