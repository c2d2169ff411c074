"""Multi-output kernels on the path (gpflow/kernels/multioutput/kernels.py:118-271)."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .. import ops
from .base import Kernel


class MultioutputKernel(Kernel):
    @property
    def num_latent_gps(self) -> int:
        raise NotImplementedError

    @property
    def latent_kernels(self):
        raise NotImplementedError

    def __call__(self, X, X2=None, *, full_cov: bool = False, full_output_cov: bool = True,
                 presliced: bool = False):
        if not presliced:
            X, X2 = self.slice(ops.to_device(X), ops.to_device(X2) if X2 is not None else None)
        if not full_cov and X2 is not None:
            raise ValueError("Ambiguous inputs: passing in `X2` is not compatible with `full_cov=False`.")
        if not full_cov:
            return self.K_diag(X, full_output_cov=full_output_cov)
        return self.K(X, X2, full_output_cov=full_output_cov)


class SharedIndependent(MultioutputKernel):
    """One kernel shared by `output_dim` independent latent GPs (kernels.py:118-197)."""

    def __init__(self, kernel: Kernel, output_dim: int):
        super().__init__()
        self.kernel = kernel
        self.output_dim = int(output_dim)

    @property
    def num_latent_gps(self) -> int:
        return self.output_dim

    @property
    def latent_kernels(self):
        return (self.kernel,)

    def K(self, X, X2=None, full_output_cov: bool = True):
        K = self.kernel.K(X, X2)  # [N, N2]
        P = self.output_dim
        if full_output_cov:
            Ks = K[:, None, :, None] * torch.eye(P, dtype=K.dtype, device=K.device)[None, :, None, :]
            return Ks  # [N, P, N2, P]
        return K[None].expand(P, *K.shape).contiguous()  # [P, N, N2]

    def K_diag(self, X, full_output_cov: bool = True):
        K = self.kernel.K_diag(X)  # [N]
        P = self.output_dim
        Ks = K[:, None].expand(-1, P)  # [N, P]
        return torch.diag_embed(Ks) if full_output_cov else Ks.contiguous()


class SeparateIndependent(MultioutputKernel):
    """One kernel per output (kernels.py:200-271)."""

    def __init__(self, kernels: Sequence[Kernel], name: Optional[str] = None):
        super().__init__(name=name)
        self.kernels = list(kernels)

    @property
    def num_latent_gps(self) -> int:
        return len(self.kernels)

    @property
    def latent_kernels(self):
        return tuple(self.kernels)

    def K(self, X, X2=None, full_output_cov: bool = True):
        Kxxs = torch.stack([k.K(X, X2) for k in self.kernels], dim=0)  # [P, N, N2]
        if full_output_cov:
            return torch.diag_embed(Kxxs.permute(1, 2, 0)).permute(0, 2, 1, 3)  # [N, P, N2, P]
        return Kxxs

    def K_diag(self, X, full_output_cov: bool = False):
        stacked = torch.stack([k.K_diag(X) for k in self.kernels], dim=1)  # [N, P]
        return torch.diag_embed(stacked) if full_output_cov else stacked
