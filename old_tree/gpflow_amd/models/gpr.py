"""GPR (gpflow/models/gpr.py:36-196)."""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops, posteriors
from ..kernels import Kernel
from ..kernels.stationaries import Stationary
from ..likelihoods import Gaussian
from ..logdensities import multivariate_normal
from ..mean_functions import MeanFunction
from .model import GPModel
from .training_mixins import InternalDataTrainingLossMixin


class GPR(GPModel, InternalDataTrainingLossMixin):
    def __init__(self, data, kernel: Kernel, mean_function: Optional[MeanFunction] = None,
                 noise_variance=None, likelihood: Optional[Gaussian] = None):
        assert (noise_variance is None) or (likelihood is None), \
            "Cannot set both `noise_variance` and `likelihood`."
        if likelihood is None:
            if noise_variance is None:
                noise_variance = 1.0
            likelihood = Gaussian(noise_variance)
        X, Y = data
        self.data = (ops.to_device(X), ops.to_device(Y))  # data_input_to_tensor, models/util.py:91-107
        if self.data[0].dim() != 2 or self.data[1].dim() != 2 or self.data[0].shape[0] != self.data[1].shape[0]:
            raise ValueError("data must be (X [N,D], Y [N,P])")
        super().__init__(kernel, likelihood, mean_function, num_latent_gps=self.data[1].shape[-1])
        self._ws = None

    def maximum_log_likelihood_objective(self):
        return self.log_marginal_likelihood()

    def log_marginal_likelihood(self) -> torch.Tensor:
        """gpr.py:91-107.  With a stationary kernel and a constant mean the whole chain
        K -> +noise -> cholesky -> triangular_solve -> reductions is ONE C-ABI call (gpk_gpr_lml);
        otherwise it is composed from the same primitives."""
        X, Y = self.data
        c = self.mean_function.constant_value()
        if isinstance(self.kernel, Stationary) and c is not None:
            Xs, _ = self.kernel.slice(X, None)
            family, var, ls = self.kernel.hyper()
            out, info = ops.gpr_lml(Xs, Y, variance=var, lengthscales=ls,
                                    noise_variance=self.likelihood.noise_for(X), mean_const=c,
                                    family=family, ws=self._ws)
            ops.check_info(info)
            return out[0]
        K = self.kernel(X)
        n = K.shape[0]
        if self.likelihood.is_heteroskedastic:   # add_likelihood_noise_cov, model_utils.py:46-50
            ops.diag_add_(K, self.likelihood.noise_for(X))
        else:
            idx = torch.arange(n, device=K.device)
            K[idx, idx] += self.likelihood.noise_variance()  # add_noise_cov, model_utils.py:33-38
        _, info = ops.potrf_(K, n, zero_upper=True)
        ops.check_info(info)
        m = self.mean_function(X)
        return multivariate_normal(Y, m, K).sum()

    def log_marginal_likelihood_and_grad(self):
        """(LML as a float, {Parameter: dLML/d(unconstrained value) as NumPy}) for the trainable parameters -- what
        `optimizers/scipy.py:322-331` obtains from TF autodiff.  SquaredExponential or Matern12 / 32 / 52 kernel (with `active_dims`), constant / zero
        mean, constant noise variance (gradients.gpr_lml_and_grad); anything else raises NotImplementedError."""
        import numpy as np
        from .. import gradients
        from ..kernels.stationaries import IsotropicStationary
        from ..mean_functions import Constant
        k, lik, mf = self.kernel, self.likelihood, self.mean_function
        c = mf.constant_value()
        from ..kernels.base import gradient_spec
        combo = gradient_spec(k, self.data[0].shape[1])   # Sum / Product of stationary kernels (kernels/base.py:216-220, 305-315)
        het = lik.is_heteroskedastic   # round 5: d LML / d sigma_n^2 per row, chained through the noise function's own reverse pass
        if combo is None and not (isinstance(k, IsotropicStationary) and k.family in ops.KERNEL_FAMILIES) or c is None \
                or not (lik.has_variance_parameter or het):
            raise NotImplementedError("gradients: SquaredExponential / Matern kernel (or a Sum / Product of them), constant mean, "
                                      "Gaussian likelihood with a variance parameter")
        X, Y = self.data
        if combo is not None:
            spec, members = combo
            lml, g, info = gradients.gpr_lml_and_grad(ops.to_device(X).contiguous(), Y, noise_variance=lik.noise_for(X),
                                                      mean_const=c, kernel_spec=spec)
            ops.check_info(info)
            gv = g["variance"].cpu().numpy()
            pairs = []
            for i, (pv, pl) in enumerate(members):
                pairs += [(pv, gv[i]), (pl, g["lengthscales"][i].cpu().numpy())]
            host = {"mean_const": g["mean_const"].cpu().numpy()}
            noise_pairs = [(lik.variance, g["noise_variance"].cpu().numpy())] if not het else \
                [(par, gv.cpu().numpy()) for par, gv in lik.noise_param_grads(X, g["noise_variance"])]
            pairs += noise_pairs
        else:
            Xs, _ = k.slice(X, None)    # active_dims (kernels/base.py:90-109); nothing is differentiated w.r.t. X
            family, var, ls = k.hyper()
            lml, g, info = gradients.gpr_lml_and_grad(Xs.contiguous(), Y, variance=var, lengthscales=ls,
                                                      noise_variance=lik.noise_for(X), mean_const=c, family=family)
            ops.check_info(info)
            host = {n: t.cpu().numpy() for n, t in g.items() if n != "noise_variance"}
            pairs = [(k.variance, host["variance"]), (k.lengthscales, host["lengthscales"])]
            pairs += [(lik.variance, g["noise_variance"].cpu().numpy())] if not het else \
                [(par, gv.cpu().numpy()) for par, gv in lik.noise_param_grads(X, g["noise_variance"])]
        if isinstance(mf, Constant) and hasattr(mf, "c"):   # (Zero is a Constant without a parameter, functions.py:195-204)
            pairs.append((mf.c, host["mean_const"]))
        out = {}
        for par, gc in pairs:
            if par.trainable:
                u = par.unconstrained_variable
                gu = np.asarray(gc, dtype=np.float64).reshape(u.shape) * par.transform.forward_grad(u)
                out[par] = out[par] + gu if par in out else gu   # (a Parameter shared by several members: k + k, tied lengthscales)
        # with parameter priors this is the log POSTERIOR density and its gradient: -training_loss (model.py:56-76)
        return self._add_log_prior(float(lml.cpu()[0]), out)

    objective_and_grad = log_marginal_likelihood_and_grad   # what optimizers.Scipy calls

    def posterior(self, precompute_cache=posteriors.PrecomputeCacheType.TENSOR) -> posteriors.GPRPosterior:
        """gpr.py:146-175"""
        return posteriors.GPRPosterior(kernel=self.kernel, data=self.data, likelihood=self.likelihood,
                                       mean_function=self.mean_function,
                                       precompute_cache=posteriors._validate_precompute_cache_type(precompute_cache)
                                       if precompute_cache is not None else None)

    def predict_f(self, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        """gpr.py:178-190: fused (no-cache) prediction."""
        return self.posterior(posteriors.PrecomputeCacheType.NOCACHE).fused_predict_f(
            Xnew, full_cov=full_cov, full_output_cov=full_output_cov)
