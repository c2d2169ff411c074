"""SVGP (gpflow/models/svgp.py:37-261)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .. import config, kullback_leiblers, ops, posteriors
from ..base import Parameter, positive, triangular
from ..conditionals import conditional
from ..inducing_variables import (InducingPoints, SharedIndependentInducingVariables,
                                  inducingpoint_wrapper)
from ..kernels import Kernel, SharedIndependent
from ..kernels.stationaries import Stationary
from ..likelihoods import Gaussian, Likelihood
from ..mean_functions import MeanFunction
from .model import GPModel
from .training_mixins import ExternalDataTrainingLossMixin


class SVGP(GPModel, ExternalDataTrainingLossMixin):
    def __init__(self, kernel: Kernel, likelihood: Likelihood, inducing_variable, *,
                 mean_function: Optional[MeanFunction] = None, num_latent_gps: int = 1,
                 q_diag: bool = False, q_mu=None, q_sqrt=None, whiten: bool = True, num_data=None):
        super().__init__(kernel, likelihood, mean_function, num_latent_gps)
        self.num_data = num_data
        self.whiten = whiten
        self.inducing_variable = inducingpoint_wrapper(inducing_variable)
        num_inducing = self.inducing_variable.num_inducing
        self._init_variational_parameters(num_inducing, q_mu, q_sqrt, q_diag)
        self._ws = None

    def _init_variational_parameters(self, num_inducing, q_mu, q_sqrt, q_diag) -> None:
        """svgp.py:90-148"""
        q_mu = np.zeros((num_inducing, self.num_latent_gps)) if q_mu is None else q_mu
        self.q_mu = Parameter(q_mu)  # [M, P]
        if q_sqrt is None:
            if q_diag:
                self.q_sqrt = Parameter(np.ones((num_inducing, self.num_latent_gps)), transform=positive())
            else:
                eye = np.array([np.eye(num_inducing) for _ in range(self.num_latent_gps)])
                self.q_sqrt = Parameter(eye, transform=triangular())  # [P, M, M]
        else:
            q_sqrt = np.asarray(q_sqrt, dtype=np.float64)
            if q_diag:
                assert q_sqrt.ndim == 2
                self.num_latent_gps = q_sqrt.shape[1]
                self.q_sqrt = Parameter(q_sqrt, transform=positive())  # [M, L|P]
            else:
                assert q_sqrt.ndim == 3
                self.num_latent_gps = q_sqrt.shape[0]
                self.q_sqrt = Parameter(q_sqrt, transform=triangular())  # [L|P, M, M]

    def prior_kl(self) -> torch.Tensor:
        """svgp.py:153-156"""
        return kullback_leiblers.prior_kl(self.inducing_variable, self.kernel, self.q_mu.device_value(),
                                          self.q_sqrt.device_value(), whiten=self.whiten)

    def maximum_log_likelihood_objective(self, data):
        return self.elbo(data)

    # ---- fused device path ---------------------------------------------------------------------
    def _fused_config(self):
        """(stationary kernel, Z tensor, mean constant) when the whole ELBO shard is one C-ABI call:
        Gaussian likelihood, constant mean, and one stationary kernel shared by all latents (plain kernel +
        InducingPoints, or SharedIndependent + SharedIndependentInducingVariables); whitened or not, full or diagonal q_sqrt."""
        if not isinstance(self.likelihood, Gaussian):
            return None
        c = self.mean_function.constant_value()
        if c is None:
            return None
        k, iv = self.kernel, self.inducing_variable
        if isinstance(k, SharedIndependent) and isinstance(iv, SharedIndependentInducingVariables):
            k, iv = k.kernel, iv.inducing_variable
        if not (isinstance(k, Stationary) and isinstance(iv, InducingPoints)):
            return None
        return k, iv.Z.device_value(), c

    def _fused_separate_config(self):
        """(member kernels, Z [m, d] | [P, m, d], mean constant) when the ELBO shard of a SeparateIndependent model is one
        C-ABI call (gpk_svgp_elbo_shard_sep): whitened, Gaussian likelihood, constant mean, full q_sqrt, stationary members
        over all input columns, inducing POINTS shared by the latents or one equally sized set per latent."""
        from ..kernels import SeparateIndependent
        from ..inducing_variables import SeparateIndependentInducingVariables
        if not self.whiten or not isinstance(self.likelihood, Gaussian):
            return None
        c = self.mean_function.constant_value()
        if c is None:
            return None
        sep = self._separate_stationary_members()
        return None if sep is None else sep + (c,)

    def _separate_stationary_members(self):
        """(member kernels, Z [m, d] | [P, m, d]) for SeparateIndependent stationary members over all input columns and
        inducing points (shared, or one equally sized set per latent), full q_sqrt; else None."""
        from ..kernels import SeparateIndependent
        from ..inducing_variables import SeparateIndependentInducingVariables
        if self.q_sqrt.device_value().dim() != 3:
            return None
        k, iv = self.kernel, self.inducing_variable
        if not isinstance(k, SeparateIndependent):
            return None
        if not all(isinstance(kk, Stationary) and kk.has_default_active_dims for kk in k.kernels):
            return None
        if isinstance(iv, SharedIndependentInducingVariables) and isinstance(iv.inducing_variable, InducingPoints):
            return k.kernels, iv.inducing_variable.Z.device_value().contiguous()
        if isinstance(iv, SeparateIndependentInducingVariables) and len(iv.inducing_variable_list) == len(k.kernels) \
                and all(isinstance(v, InducingPoints) for v in iv.inducing_variable_list):
            Zs = [v.Z.device_value() for v in iv.inducing_variable_list]
            if len({tuple(z.shape) for z in Zs}) == 1:
                return k.kernels, torch.stack(Zs).contiguous()
        return None

    def _unwhitened_shared_factor_config(self):
        """(stationary kernel, Z) when the un-whitened ELBO can run on ONE factorisation of Kuu: one stationary kernel
        shared by the latents over inducing points, full q_sqrt."""
        if self.whiten or self.q_sqrt.device_value().dim() != 3:
            return None
        k, iv = self.kernel, self.inducing_variable
        if isinstance(k, SharedIndependent) and isinstance(iv, SharedIndependentInducingVariables):
            k, iv = k.kernel, iv.inducing_variable
        if not (isinstance(k, Stationary) and isinstance(iv, InducingPoints)):
            return None
        return k, iv.Z.device_value()

    def _elbo_terms_unwhitened(self, X, Y, k, Z):
        """whiten=False on one trapezoid.  The reference factors Kuu twice per ELBO -- `prior_kl` -> `gauss_kl(K=Kuu)`
        (kullback_leiblers.py:107) and the conditional (conditionals/util.py:67) -- and so did the composed path here: two
        latency chains of 16 panels each (profiles/r04_unwhitened_timeline_before.txt).  Here [Kuu + jitter I ; Kfu ; q_mu^T ;
        tril(q_sqrt_p)^T] goes through ONE factorisation: the minibatch rows come back as A^T = Kfu Lm^-T (util.py:125), the
        others as (Lm^-1 q_mu)^T and (Lm^-1 Lq_p)^T, i.e. the Mahalanobis and trace terms of the KL (:114, :152) -- and, read as
        the whitened parameters of the same q, everything the conditional needs without its second triangular solve (below).
        Cm shape: 4.45 ms (two factorisations) -> 3.72 (one) -> 2.63 ms (no Lm^-T solve of the minibatch rows)."""
        Xs, Zs = k.slice(X, Z)
        q_mu, q_sqrt = self.q_mu.device_value(), self.q_sqrt.device_value()
        M, P = q_mu.shape
        B = Xs.shape[0]
        T = torch.empty((M + B + P + P * M, M), dtype=torch.float64, device=Xs.device)
        k.K_into(Zs, None, T[:M], diag_add=config.default_jitter(), lower_only=True)
        if B:
            k.K_into(Xs, Zs, T[M:M + B])
        T[M + B:M + B + P] = q_mu.t()
        ops.transpose(q_sqrt.contiguous(), mode=1, out=T[M + B + P:].view(P, M, M))       # tril(q_sqrt_p)^T
        _, info = ops.potrf_(T, M)
        # KL[q || N(0, Kuu)]  (kullback_leiblers.py:98-165)
        mahalanobis = ops.sumsq(T[M + B:M + B + P])[0]
        trace = ops.sumsq(T[M + B + P:])[0]
        logdet_qcov = torch.log(torch.diagonal(q_sqrt, dim1=-2, dim2=-1) ** 2).sum()
        kl = 0.5 * (mahalanobis - float(M * P) - logdet_qcov + trace + float(P) * 2.0 * ops.sum_log_diag(T[:M])[0])
        # q(f) at the minibatch (posteriors.py:828-841 -> conditionals/util.py:128-167).  The un-whitened q(u) = N(q_mu, Lq Lq^T)
        # IS the whitened q(v), v = Lm^-1 u, with mean Lm^-1 q_mu and square root G_p = Lm^-1 Lq_p (lower triangular again) --
        # exactly the two blocks of rows the KL needed: fmean = A^T (Lm^-1 q_mu)  and  sum_j (Lq^T Lm^-T A)_j^2 = sum_j (G^T A)_j^2
        # (util.py:139-164 with the Lm^-T solve of the N columns of A folded into the M x M factor).  No second triangular
        # solve of the minibatch rows: M^2 B flop and a 1-ms chain of launches less than the literal form.
        At = T[M:M + B]
        V = T[M + B:M + B + P].t().contiguous()                   # Lm^-1 q_mu  [M, P]
        GT = T[M + B + P:].view(P, M, M)                          # G_p^T (upper): the LqT operand of the projection
        s0, f_mean, _ = ops.row_stats(At, V=V)
        ssq = ops.project(At, GT)
        f_var = (k.K_diag(Xs)[None, :] - s0[None, :] + ssq).t().contiguous()
        f_mean = f_mean + self.mean_function(X)
        var_exp = self.likelihood.variational_expectations(X, f_mean, f_var, Y)
        ops.check_info(info)
        return torch.stack([var_exp.sum(), kl])

    def _elbo_terms_unwhitened_separate(self, X, Y, kernels, Z):
        """`_elbo_terms_unwhitened` for one kernel PER latent (SeparateIndependent): the P trapezoids [Kuu_p ; Kfu_p ; q_mu_p^T ;
        tril(q_sqrt_p)^T] go through ONE batched factorisation; per latent the extra rows are A_p^T, (Lm_p^-1 q_mu_p)^T and
        G_p^T = (Lm_p^-1 Lq_p)^T -- KL terms and whitened parameters of the same q at once (kullback_leiblers.py:98-165 batched
        over K [L,M,M]; conditionals/util.py:566-629 with white = False)."""
        q_mu, q_sqrt = self.q_mu.device_value(), self.q_sqrt.device_value()
        M, P = q_mu.shape
        B = X.shape[0]
        T = torch.empty((P, M + B + 1 + M, M), dtype=torch.float64, device=X.device)
        for p, kp in enumerate(kernels):
            Zp = Z if Z.dim() == 2 else Z[p]
            kp.K_into(Zp, None, T[p, :M], diag_add=config.default_jitter(), lower_only=True)
            if B:
                kp.K_into(X, Zp, T[p, M:M + B])
        T[:, M + B] = q_mu.t()
        ops.transpose(q_sqrt.contiguous(), mode=1, out=T[:, M + B + 1:])                 # tril(q_sqrt_p)^T
        _, info = ops.potrf_(T, M)
        arow = T[:, M + B].contiguous()                                                  # [P, M]: (Lm_p^-1 q_mu_p)^T
        GT = T[:, M + B + 1:].contiguous()                                               # [P, M, M]: G_p^T (upper)
        mahalanobis = ops.sumsq(arow)[0]
        trace = ops.sumsq(GT.reshape(P * M, M))[0]
        logdet_qcov = torch.log(torch.diagonal(q_sqrt, dim1=-2, dim2=-1) ** 2).sum()
        kl = 0.5 * (mahalanobis - float(M * P) - logdet_qcov + trace + 2.0 * ops.sum_log_diag(T[:, :M]).sum())
        s0s, mus = [], []
        for p in range(P):
            s0, mu, _ = ops.row_stats(T[p, M:M + B], V=arow[p].reshape(M, 1).contiguous())
            s0s.append(s0)
            mus.append(mu[:, 0])
        ssq = ops.project(T[:, M:M + B], GT)                                             # [P, B]: sum_j (G_p^T A_p)_j^2
        kdiag = torch.stack([kp.K_diag(X) for kp in kernels])                            # [P, B]
        f_var = (kdiag - torch.stack(s0s) + ssq).t().contiguous()
        f_mean = torch.stack(mus, dim=-1) + self.mean_function(X)
        var_exp = self.likelihood.variational_expectations(X, f_mean, f_var, Y)
        ops.check_info(info)
        return torch.stack([var_exp.sum(), kl])

    def elbo_terms(self, data):
        """(sum_b var_exp_b over the given rows, KL) as a 2-element device tensor -- the two pieces
        svgp.py:172-174 combines; the first is what gets all-reduced when the minibatch is sharded."""
        X, Y = ops.to_device(data[0]), ops.to_device(data[1])
        fused = self._fused_config()
        if fused is not None:
            k, Z, c = fused
            Xs, Zs = k.slice(X, Z)
            family, var, ls = k.hyper()
            m, rows, d, P = Zs.shape[0], Xs.shape[0], Zs.shape[1], self.q_mu.shape[1]
            q_sqrt = self.q_sqrt.device_value()
            key = (m, rows, d, P, q_sqrt.dim() == 2, bool(self.whiten))
            if self._ws is None or self._ws[0] != key:
                self._ws = (key, ops.svgp_elbo_workspace(m, rows, d, P, q_sqrt.dim() == 2, self.whiten))
            out, info = ops.svgp_elbo_shard(Zs, Xs, Y, self.q_mu.device_value(), q_sqrt, variance=var,
                                            lengthscales=ls, noise_variance=self.likelihood.noise_for(X),
                                            jitter=config.default_jitter(), mean_const=c, family=family,
                                            ws=self._ws[1], whiten=self.whiten)
            ops.check_info(info)
            return out
        sep = self._fused_separate_config()
        if sep is not None:
            kernels, Zs, c = sep
            hyp = [k.hyper() for k in kernels]
            P, m, d, rows = len(kernels), Zs.shape[-2], Zs.shape[-1], X.shape[0]
            ls = [np.atleast_1d(h[2]) for h in hyp]
            if any(l.size > 1 for l in ls):   # mixed isotropic / ARD members: every row spelled out
                ls = np.stack([np.broadcast_to(l, (d,)) for l in ls])
            else:
                ls = np.concatenate(ls)
            key = ("sep", m, rows, d, P)
            if self._ws is None or self._ws[0] != key:
                self._ws = (key, ops.svgp_elbo_sep_workspace(m, rows, d, P))
            out, info = ops.svgp_elbo_shard_sep(Zs, X.contiguous(), Y, self.q_mu.device_value(), self.q_sqrt.device_value(),
                                                variances=[h[1] for h in hyp], lengthscales=ls, families=[h[0] for h in hyp],
                                                noise_variance=self.likelihood.noise_for(X),
                                                jitter=config.default_jitter(), mean_const=c, ws=self._ws[1])
            ops.check_info(info)
            return out
        shared = self._unwhitened_shared_factor_config()
        if shared is not None:
            return self._elbo_terms_unwhitened(X, Y, *shared)
        if not self.whiten:
            members = self._separate_stationary_members()
            if members is not None:
                return self._elbo_terms_unwhitened_separate(X, Y, *members)
        kl = self.prior_kl()
        f_mean, f_var = self.predict_f(X, full_cov=False, full_output_cov=False)
        var_exp = self.likelihood.variational_expectations(X, f_mean, f_var, Y)
        return torch.stack([var_exp.sum(), kl])

    def elbo(self, data) -> torch.Tensor:
        """svgp.py:166-181"""
        X = data[0]
        terms = self.elbo_terms(data)
        if self.num_data is not None:
            scale = float(self.num_data) / float(X.shape[0])
        else:
            scale = 1.0
        return terms[0] * scale - terms[1]

    def gradient_config(self, allow_active_dims: bool = False, allow_q_diag: bool = False, allow_heteroskedastic: bool = False):
        """(kernel, InducingPoints, mean constant) if the hand-written reverse pass covers this model: whitened or not,
        Gaussian likelihood with a variance parameter, full q_sqrt, constant mean, and ONE isotropic stationary kernel
        (SquaredExponential / Matern12 / 32 / 52; `active_dims` and `q_diag` only where the caller
        handles them itself: `elbo_and_grad`) over InducingPoints -- either directly or as SharedIndependent +
        SharedIndependentInducingVariables (BASELINE config C5: P latents share Kuu / Kuf).  SeparateIndependent kernels
        are differentiated latent by latent (`_separate_gradient_config`).  Raises NotImplementedError."""
        from ..kernels.stationaries import IsotropicStationary
        k, iv, lik = self.kernel, self.inducing_variable, self.likelihood
        if isinstance(k, SharedIndependent) and isinstance(iv, SharedIndependentInducingVariables):
            k, iv = k.kernel, iv.inducing_variable
        c = self.mean_function.constant_value()
        noise_ok = isinstance(lik, Gaussian) and (lik.has_variance_parameter or (allow_heteroskedastic and lik.is_heteroskedastic))
        if not (isinstance(k, IsotropicStationary) and k.family in ops.KERNEL_FAMILIES and noise_ok
                and isinstance(iv, InducingPoints) and c is not None
                and (self.q_sqrt.numpy().ndim == 3 or (allow_q_diag and self.q_sqrt.numpy().ndim == 2))
                and (allow_active_dims or k.has_default_active_dims)):
            raise NotImplementedError("gradients: SVGP with a SquaredExponential / Matern kernel (optionally shared by independent "
                                      "latents, or one per latent), Gaussian likelihood, InducingPoints, full q_sqrt, constant mean")
        return k, iv, float(c)

    def _separate_gradient_config(self):
        """[(isotropic stationary kernel_p, InducingPoints_p)] per latent for SeparateIndependent kernels (over shared or
        separate inducing points), else None."""
        from ..covariances import _pairs  # noqa: F401  (same pairing rule as Kuu / Kuf)
        from ..kernels import SeparateIndependent
        from ..kernels.stationaries import IsotropicStationary
        from ..inducing_variables import SeparateIndependentInducingVariables
        k, iv, lik = self.kernel, self.inducing_variable, self.likelihood
        if not isinstance(k, SeparateIndependent):
            return None
        if isinstance(iv, SeparateIndependentInducingVariables):
            ivs = list(iv.inducing_variable_list)
        elif isinstance(iv, SharedIndependentInducingVariables):
            ivs = [iv.inducing_variable] * len(k.kernels)
        else:
            return None
        c = self.mean_function.constant_value()
        if not (all(isinstance(kk, IsotropicStationary) and kk.family in ops.KERNEL_FAMILIES for kk in k.kernels) and all(isinstance(v, InducingPoints) for v in ivs)
                and isinstance(lik, Gaussian) and (lik.has_variance_parameter or lik.is_heteroskedastic) and c is not None
                and self.q_sqrt.numpy().ndim == 3
                and len(ivs) == len(k.kernels)):
            raise NotImplementedError("gradients: SeparateIndependent needs SquaredExponential / Matern members over InducingPoints, a "
                                      "Gaussian likelihood, full q_sqrt and a constant mean")
        return list(zip(k.kernels, ivs)), float(c)

    @staticmethod
    def _sliced(k, Z, X):
        """Inputs restricted to the kernel's active_dims + the scatter of a gradient w.r.t. the sliced Z back to Z's shape
        (gpflow/kernels/base.py:90-109: the kernel only ever sees these columns, so dF/dZ is zero elsewhere)."""
        if k.has_default_active_dims:
            return Z, X, (lambda gz: gz)
        Xs, Zs = k.slice(X, Z)
        dims = k._active_dims
        cols = torch.arange(Z.shape[1], device=Z.device)[dims] if isinstance(dims, slice) else torch.as_tensor(dims, device=Z.device)

        def scatter(gz):
            full = torch.zeros_like(Z)
            full.index_add_(1, cols, gz)   # (a repeated active column collects both contributions, like tf.gather's gradient)
            return full
        return Zs, Xs, scatter

    def elbo_and_grad(self, data):
        """(ELBO on `data` as a float, {Parameter: dELBO/d(unconstrained value) as NumPy}) for the trainable parameters
        -- the pair `optimizers/scipy.py:322-331` gets from TF autodiff over `training_loss_closure(data)`.  Whitened or
        not, SquaredExponential or Matern12 / 32 / 52 kernel (with `active_dims`; shared by the latents or one per latent), Gaussian likelihood,
        InducingPoints, full q_sqrt (gradients.svgp_elbo_and_grad).  For minibatch training keep the variables on the
        device instead: training.SVGPTrainer."""
        from .. import gradients
        from ..base import FillTriangular
        from ..mean_functions import Constant
        lik, mf = self.likelihood, self.mean_function
        # scope checks first: a model outside the reverse pass is refused before anything touches the device
        sep = self._separate_gradient_config()
        from ..kernels.base import gradient_spec
        combo = gradient_spec(self.kernel, int(tuple(data[0].shape)[-1])) if sep is None else None   # Sum / Product of stationary kernels
        if combo is not None:
            return self._elbo_and_grad_combination(data, combo)
        # (a heteroskedastic Gaussian likelihood -- per-row dF/d sigma_n^2 chained through the noise function -- in the single-kernel
        #  reverse passes, whitened and un-whitened: round 5)
        single = self.gradient_config(allow_active_dims=True, allow_q_diag=True, allow_heteroskedastic=True) if sep is None else None
        X, Y = ops.to_device(data[0]), ops.to_device(data[1])
        scale = 1.0 if self.num_data is None else float(self.num_data) / float(X.shape[0])
        fn = gradients.svgp_elbo_and_grad if self.whiten else gradients.svgp_elbo_and_grad_unwhitened
        het = lik.is_heteroskedastic
        common = dict(noise_variance=lik.noise_for(X), jitter=config.default_jitter(), scale=scale)
        pairs = []
        if sep is None:
            k, iv, c = single
            Zs, Xs, scatter = self._sliced(k, iv.Z.device_value(), X)
            family, var, ls = k.hyper()
            F, g, info = fn(Zs, Xs, Y, self.q_mu.device_value(), self.q_sqrt.device_value(), variance=var, lengthscales=ls,
                            mean_const=float(c), family=family, **common)
            ops.check_info(info)
            Fv = float(F.cpu()[0])
            host = {n: (scatter(t) if n == "Z" else t).cpu().numpy() for n, t in g.items()}
            pairs = [(k.variance, host["variance"]), (k.lengthscales, host["lengthscales"]), (iv.Z, host["Z"])]
            g_noise, g_mean, g_qmu, g_qs = host["noise_variance"], host["mean_const"], host["q_mu"], host["q_sqrt"]
            if het:
                pairs += [(par, gv.cpu().numpy()) for par, gv in lik.noise_param_grads(X, g["noise_variance"])]
        else:
            # SeparateIndependent (conditionals/util.py:566-629): L independent single-output problems that share the
            # likelihood, the mean constant and the rows of the minibatch; ELBO and the shared gradients are their sums
            members, c = sep
            q_mu, q_sqrt = self.q_mu.device_value(), self.q_sqrt.device_value()
            Fv, g_noise, g_mean = 0.0, 0.0, 0.0
            g_qmu = np.zeros(tuple(q_mu.shape))
            g_qs = np.zeros(tuple(q_sqrt.shape))
            zgrads = {}
            for p_, (k, iv) in enumerate(members):
                Zs, Xs, scatter = self._sliced(k, iv.Z.device_value(), X)
                family, var, ls = k.hyper()
                F, g, info = fn(Zs, Xs, Y[:, p_:p_ + 1].contiguous(), q_mu[:, p_:p_ + 1].contiguous(), q_sqrt[p_:p_ + 1].contiguous(),
                                variance=var, lengthscales=ls, mean_const=float(c), family=family, **common)
                ops.check_info(info)
                Fv += float(F.cpu()[0])
                host = {n: (scatter(t) if n == "Z" else t).cpu().numpy() for n, t in g.items()}
                pairs += [(k.variance, host["variance"]), (k.lengthscales, host["lengthscales"])]
                zgrads[id(iv.Z)] = (iv.Z, zgrads.get(id(iv.Z), (None, 0.0))[1] + host["Z"])  # shared Z: contributions add up
                g_noise = g_noise + host["noise_variance"]
                g_mean = g_mean + host["mean_const"]
                g_qmu[:, p_:p_ + 1] = host["q_mu"]
                g_qs[p_:p_ + 1] = host["q_sqrt"]
            pairs += list(zgrads.values())
            if het:   # the latents share the likelihood: their per-row dF/d sigma_n^2 add up before the noise Function's reverse pass
                pairs += [(par, gv.cpu().numpy()) for par, gv in lik.noise_param_grads(X, ops.to_device(np.asarray(g_noise)))]
        if not het:
            pairs.append((lik.variance, g_noise))
        pairs += [(self.q_mu, g_qmu), (self.q_sqrt, g_qs)]
        if isinstance(mf, Constant) and hasattr(mf, "c"):   # (Zero is a Constant without a parameter, functions.py:195-204)
            pairs.append((mf.c, g_mean))
        out = {}
        for par, gc in pairs:
            if not par.trainable:
                continue
            u = par.unconstrained_variable
            if isinstance(par.transform, FillTriangular):   # linear embedding: the vector entries are the lower-triangular ones
                gu = par.transform.inverse(np.asarray(gc, dtype=np.float64)).reshape(u.shape)
            else:
                gu = np.asarray(gc, dtype=np.float64).reshape(u.shape) * par.transform.forward_grad(u)
            out[par] = out[par] + gu if par in out else gu
        return self._add_log_prior(Fv, out)   # (+ log prior density of the trainable parameters: model.py:56-76)

    def _elbo_and_grad_combination(self, data, combo):
        """elbo_and_grad for a Sum / Product of stationary kernels (kernels/base.py:216-220, 305-315), members possibly over
        different `active_dims`: the whitened or un-whitened reverse pass with the members' adjoints taken one by one
        (gradients.KernelSpec)."""
        from .. import gradients
        from ..base import FillTriangular
        from ..mean_functions import Constant
        spec, members = combo
        lik, mf, iv = self.likelihood, self.mean_function, self.inducing_variable
        c = mf.constant_value()
        # (a diagonal q_sqrt [M, P] goes through the same two reverse passes: the covariance spec and the q_diag branches of
        #  gradients.svgp_elbo_and_grad / _unwhitened are independent of each other)
        het = isinstance(lik, Gaussian) and lik.is_heteroskedastic   # (per-row dF/d sigma_n^2 chained through the noise Function)
        if not (isinstance(lik, Gaussian) and (lik.has_variance_parameter or het) and isinstance(iv, InducingPoints) and c is not None):
            raise NotImplementedError("gradients with a kernel combination: Gaussian likelihood (a variance parameter or a noise "
                                      "Function), InducingPoints, constant mean")
        X, Y = ops.to_device(data[0]), ops.to_device(data[1])
        scale = 1.0 if self.num_data is None else float(self.num_data) / float(X.shape[0])
        fn = gradients.svgp_elbo_and_grad if self.whiten else gradients.svgp_elbo_and_grad_unwhitened
        F, g, info = fn(iv.Z.device_value(), X.contiguous(), Y, self.q_mu.device_value(), self.q_sqrt.device_value(),
                        noise_variance=lik.noise_for(X), jitter=config.default_jitter(), scale=scale,
                        mean_const=float(c), kernel_spec=spec)
        ops.check_info(info)
        gv = g["variance"].cpu().numpy()
        pairs = []
        for i, (pv, pl) in enumerate(members):
            pairs += [(pv, gv[i]), (pl, g["lengthscales"][i].cpu().numpy())]
        pairs += [(iv.Z, g["Z"].cpu().numpy()), (self.q_mu, g["q_mu"].cpu().numpy()), (self.q_sqrt, g["q_sqrt"].cpu().numpy())]
        pairs += [(par, gv.cpu().numpy()) for par, gv in lik.noise_param_grads(X, g["noise_variance"])] if het else \
            [(lik.variance, g["noise_variance"].cpu().numpy())]
        if isinstance(mf, Constant) and hasattr(mf, "c"):   # (Zero is a Constant without a parameter, functions.py:195-204)
            pairs.append((mf.c, g["mean_const"].cpu().numpy()))
        out = {}
        for par, gc in pairs:
            if not par.trainable:
                continue
            u = par.unconstrained_variable
            if isinstance(par.transform, FillTriangular):
                gu = par.transform.inverse(np.asarray(gc, dtype=np.float64)).reshape(u.shape)
            else:
                gu = np.asarray(gc, dtype=np.float64).reshape(u.shape) * par.transform.forward_grad(u)
            out[par] = out[par] + gu if par in out else gu
        return self._add_log_prior(float(F.cpu()[0]), out)

    def posterior(self, precompute_cache=posteriors.PrecomputeCacheType.TENSOR):
        """svgp.py:210-240"""
        return posteriors.create_posterior(self.kernel, self.inducing_variable, self.q_mu, self.q_sqrt,
                                           whiten=self.whiten, mean_function=self.mean_function,
                                           precompute_cache=precompute_cache)

    def predict_f(self, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        """svgp.py:243-255"""
        return self.posterior(posteriors.PrecomputeCacheType.NOCACHE).fused_predict_f(
            Xnew, full_cov=full_cov, full_output_cov=full_output_cov)
