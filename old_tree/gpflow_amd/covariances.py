"""Kuu / Kuf (gpflow/covariances/{kuus,kufs}.py, multioutput/{kuus,kufs}.py) for the inducing-point x
stationary-kernel pairs on the path.  The reference dispatches on (inducing type, kernel type) with
multipledispatch; here the same table is a plain isinstance ladder."""
from __future__ import annotations

import torch

from . import ops
from .inducing_variables import (InducingPoints, SeparateIndependentInducingVariables,
                                 SharedIndependentInducingVariables)
from .kernels import Kernel, MultioutputKernel, SeparateIndependent, SharedIndependent


def _pairs(inducing_variable, kernel):
    """(Z tensor, latent kernel) per latent GP for the Separate* combinations
    (multioutput/kuus.py:65-121, kufs.py:63-115)."""
    if isinstance(kernel, SeparateIndependent):
        ks = list(kernel.kernels)
    elif isinstance(kernel, SharedIndependent):
        ks = None
    else:
        raise NotImplementedError(f"Kuu/Kuf: unsupported kernel {type(kernel).__name__}")
    if isinstance(inducing_variable, SeparateIndependentInducingVariables):
        zs = [iv.Z.device_value() for iv in inducing_variable.inducing_variable_list]
    elif isinstance(inducing_variable, SharedIndependentInducingVariables):
        zs = None
    else:
        raise NotImplementedError(f"Kuu/Kuf: unsupported inducing variable {type(inducing_variable).__name__}")
    L = len(ks) if ks is not None else len(zs)
    if ks is None:
        ks = [kernel.kernel] * L
    if zs is None:
        zs = [inducing_variable.inducing_variable.Z.device_value()] * L
    if len(ks) != len(zs):
        raise ValueError("number of kernels and of inducing-variable sets differ")
    return list(zip(zs, ks))


def Kuu(inducing_variable, kernel, *, jitter: float = 0.0) -> torch.Tensor:
    """[M,M] (InducingPoints x Kernel: kuus.py:24-34; Shared x Shared: multioutput/kuus.py:49-62)
    or [L,M,M] (any Separate combination: multioutput/kuus.py:65-121)."""
    if isinstance(inducing_variable, InducingPoints) and not isinstance(kernel, MultioutputKernel):
        Z = inducing_variable.Z.device_value()
        Zs, _ = kernel.slice(Z, None)
        return kernel.K_into(Zs, None, None, diag_add=jitter)
    if isinstance(inducing_variable, SharedIndependentInducingVariables) and isinstance(kernel, SharedIndependent):
        return Kuu(inducing_variable.inducing_variable, kernel.kernel, jitter=jitter)
    pairs = _pairs(inducing_variable, kernel)
    return torch.stack([k.K_into(k.slice(z, None)[0], None, None, diag_add=jitter) for z, k in pairs], dim=0)


def Kfu(inducing_variable, kernel, Xnew) -> torch.Tensor:
    """K(Xnew, Z): the row-major [N,M] (or [L,N,M]) layout the device solves consume."""
    Xnew = ops.to_device(Xnew)
    if isinstance(inducing_variable, InducingPoints) and not isinstance(kernel, MultioutputKernel):
        Z = inducing_variable.Z.device_value()
        Xs, Zs = kernel.slice(Xnew, Z)
        return kernel.K_into(Xs, Zs, None)
    if isinstance(inducing_variable, SharedIndependentInducingVariables) and isinstance(kernel, SharedIndependent):
        return Kfu(inducing_variable.inducing_variable, kernel.kernel, Xnew)
    pairs = _pairs(inducing_variable, kernel)
    outs = []
    for z, k in pairs:
        Xs, Zs = k.slice(Xnew, z)
        outs.append(k.K_into(Xs, Zs, None))
    return torch.stack(outs, dim=0)


def Kuf(inducing_variable, kernel, Xnew) -> torch.Tensor:
    """[M,N] (kufs.py:25-34, multioutput/kufs.py:49-60) or [L,M,N] (kufs.py:63-115) -- returned as a
    transposed view of the [N,M] buffer built by Kfu."""
    return Kfu(inducing_variable, kernel, Xnew).transpose(-1, -2)
