"""Log densities on the path (gpflow/logdensities.py:29-30, 139-156)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops

LOG2PI = float(np.log(2 * np.pi))


def gaussian(x: torch.Tensor, mu: torch.Tensor, var) -> torch.Tensor:
    """logdensities.py:29-30 (elementwise glue)"""
    var_t = var if isinstance(var, torch.Tensor) else torch.as_tensor(float(var), dtype=torch.float64, device=x.device)
    return -0.5 * (LOG2PI + torch.log(var_t) + (mu - x) ** 2 / var_t)


def multivariate_normal(x, mu, L) -> torch.Tensor:
    """logdensities.py:139-156: x, mu [D, N] (N broadcastable), L [D, D] lower -> [N].
    alpha = L^-1 (x - mu) is a blocked MFMA solve (gpk_trsm), the rest two device reductions."""
    x, mu, L = ops.to_device(x), ops.to_device(mu), ops.to_device(L)
    d = x - mu  # [D, N]
    D, N = d.shape
    dT = ops.transpose(d.contiguous())  # [N, D]: rows are right-hand sides
    invd = ops.trtri_blocks(L)
    ops.trsm_(dT, L, invd, trans=0)  # alpha^T
    ss, _, _ = ops.row_stats(dT)  # [N] sum_i alpha_i^2
    logdet = ops.sum_log_diag(L)  # [1]
    return -0.5 * ss - 0.5 * D * LOG2PI - logdet
