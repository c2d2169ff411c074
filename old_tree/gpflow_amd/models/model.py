"""Model bases (gpflow/models/model.py:31-343)."""
from __future__ import annotations

import abc
from typing import Optional

import torch

from .. import ops
from ..base import Module
from ..config import default_jitter
from ..kernels import Kernel, MultioutputKernel
from ..likelihoods import Likelihood
from ..mean_functions import MeanFunction, Zero
from ..posteriors import assert_params_false


class BayesianModel(Module, metaclass=abc.ABCMeta):
    def log_prior_density(self) -> float:
        """model.py:47-54"""
        return float(sum(p.log_prior_density() for p in self.trainable_parameters))

    def _add_log_prior(self, value: float, grads: dict):
        """MAP estimation (model.py:47-76): every training loss is -(objective + log_prior_density) with the priors of ALL
        trainable parameters; the device reverse pass returns d objective / d(unconstrained), this adds the prior part
        (Parameter.log_prior_density_grad, host arithmetic on a few scalars).  `grads` is {Parameter: gradient}."""
        missing = [q for q in self.trainable_parameters if q.prior is not None and q not in grads]
        if missing:
            raise NotImplementedError(f"a trainable parameter with a prior is outside this model's reverse pass: {missing[0]!r}")
        for par in grads:
            if par.prior is not None:
                grads[par] = grads[par] + par.log_prior_density_grad()
        return value + self.log_prior_density(), grads

    def log_posterior_density(self, *args, **kwargs):
        return self.maximum_log_likelihood_objective(*args, **kwargs) + self.log_prior_density()

    def _training_loss(self, *args, **kwargs):
        """model.py:71-76"""
        return -(self.maximum_log_likelihood_objective(*args, **kwargs) + self.log_prior_density())

    @abc.abstractmethod
    def maximum_log_likelihood_objective(self, *args, **kwargs):
        raise NotImplementedError


class GPModel(BayesianModel):
    """model.py:96-343"""

    def __init__(self, kernel: Kernel, likelihood: Likelihood, mean_function: Optional[MeanFunction] = None,
                 num_latent_gps: Optional[int] = None):
        assert num_latent_gps is not None, "GPModel requires specification of num_latent_gps"
        self.num_latent_gps = num_latent_gps
        if mean_function is None:
            mean_function = Zero()
        self.mean_function = mean_function
        self.kernel = kernel
        self.likelihood = likelihood

    @staticmethod
    def calc_num_latent_gps_from_data(data, kernel, likelihood) -> int:
        """model.py:142-151"""
        _, Y = data
        output_dim = Y.shape[-1]
        if isinstance(kernel, MultioutputKernel):
            return kernel.num_latent_gps
        return output_dim

    @abc.abstractmethod
    def predict_f(self, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        raise NotImplementedError

    def predict_f_samples(self, Xnew, num_samples: Optional[int] = None, full_cov: bool = True,
                          full_output_cov: bool = False) -> torch.Tensor:
        """model.py:232-280 (sampling itself is downstream of the hot path: 'next' row of the scope
        table).  Draws use the device Cholesky of the predictive covariance."""
        if full_cov and full_output_cov:
            raise NotImplementedError(
                "The combination of both `full_cov` and `full_output_cov` is not supported.")
        mean, cov = self.predict_f(Xnew, full_cov=full_cov, full_output_cov=full_output_cov)
        S = 1 if num_samples is None else num_samples
        if full_cov:  # cov [P,N,N]
            P, N, _ = cov.shape
            T = cov.clone()
            idx = torch.arange(N, device=cov.device)
            T[:, idx, idx] += default_jitter()  # conditionals/util.py:196-201
            _, info = ops.potrf_(T, N, zero_upper=True)
            ops.check_info(info)
            epsT = torch.randn((P, S, N), dtype=torch.float64, device=cov.device)
            zT = ops.gemm_nt(epsT, T)  # [P,S,N]: (L eps)^T on the fp64 MFMA GEMM
            samples = (mean.t()[:, None, :] + zT).permute(1, 2, 0).contiguous()  # [S,N,P]
        else:
            eps = torch.randn((S,) + tuple(mean.shape), dtype=torch.float64, device=mean.device)
            samples = mean[None] + eps * torch.sqrt(cov)[None]  # (no clamp: conditionals/util.py:193 takes tf.sqrt(cov) as is)
        return samples[0] if num_samples is None else samples

    def predict_y(self, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        """model.py:290-325"""
        assert_params_false(self.predict_y, full_cov=full_cov, full_output_cov=full_output_cov)
        f_mean, f_var = self.predict_f(Xnew, full_cov=full_cov, full_output_cov=full_output_cov)
        return self.likelihood.predict_mean_and_var(Xnew, f_mean, f_var)

    def predict_log_density(self, data, full_cov: bool = False, full_output_cov: bool = False):
        """model.py:327-343"""
        assert_params_false(self.predict_log_density, full_cov=full_cov, full_output_cov=full_output_cov)
        X, Y = data
        f_mean, f_var = self.predict_f(X, full_cov=full_cov, full_output_cov=full_output_cov)
        return self.likelihood.predict_log_density(X, f_mean, f_var, ops.to_device(Y))
