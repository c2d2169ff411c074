"""Posterior objects (gpflow/posteriors.py:97-114, 193-443, 640-887, 1039-1108): fused (no cache)
and cached prediction for GPR and for the independent single-/multi-output SVGP posteriors."""
from __future__ import annotations

import enum
from abc import ABC, abstractmethod
from typing import Optional, Tuple, Type, Union

import torch

from . import config, covariances, ops
from .base import Module
from .conditionals import (Factor, base_conditional, conditional_tail, expand_independent_outputs,
                           factor_with_rows, separate_independent_trapezoid_tail, tail_over_batches)
from .inducing_variables import (InducingPoints, InducingVariables,
                                 SeparateIndependentInducingVariables,
                                 SharedIndependentInducingVariables)
from .kernels import Kernel, MultioutputKernel, SeparateIndependent, SharedIndependent


class PrecomputeCacheType(enum.Enum):
    """posteriors.py:97-114"""
    TENSOR = "tensor"
    VARIABLE = "variable"
    NOCACHE = "nocache"


def _validate_precompute_cache_type(value) -> PrecomputeCacheType:
    """posteriors.py:146-160"""
    if value is None:
        return PrecomputeCacheType.NOCACHE
    elif isinstance(value, PrecomputeCacheType):
        return value
    elif isinstance(value, str):
        return PrecomputeCacheType(value.lower())
    else:
        raise ValueError(
            f"{value} is not a valid PrecomputeCacheType."
            " Valid options: 'tensor', 'variable', 'nocache' (or None).")


def assert_params_false(called_method, **kwargs: bool) -> None:
    """gpflow/utilities/model_utils.py:10-25"""
    true_kwargs = {k for k, v in kwargs.items() if v}
    if true_kwargs:
        raise NotImplementedError(
            f"{called_method.__name__} does not currently support: {' and '.join(sorted(true_kwargs))}")


def _flatten_rows(Xnew: torch.Tensor):
    """[..., T, D] -> ([prod*T, D], leading shape) (posteriors handle leading batch dims by
    broadcasting, conditionals/util.py:108-124; here they become rows)."""
    lead = tuple(Xnew.shape[:-1])
    return Xnew.reshape(-1, Xnew.shape[-1]).contiguous(), lead


def _knn_of(kernel_call, Xf: torch.Tensor, lead, full_cov: bool):
    """Knn accessor for tail_over_batches: marginal variances of all rows at once, or the [T, T] block of one batch
    element of Xnew [batch..., T, D] (the reference's kernel broadcasts over the batch dims, util.py:108-131)."""
    if full_cov and len(lead) > 1:
        T = lead[-1]
        return lambda b: kernel_call(Xf[b * T:(b + 1) * T], True)
    cache = {}

    def all_rows(b):
        if "k" not in cache:
            cache["k"] = kernel_call(Xf, full_cov)
        return cache["k"]
    return all_rows


class AbstractPosterior(Module, ABC):
    """posteriors.py:193-358"""

    def __init__(self, kernel, X_data, cache=None, mean_function=None) -> None:
        self.kernel = kernel
        self.X_data = X_data
        self.cache = cache
        self.mean_function = mean_function
        self._precompute_cache: Optional[PrecomputeCacheType] = None

    def _add_mean_function(self, Xnew, mean):
        if self.mean_function is None:
            return mean
        return mean + self.mean_function(Xnew)

    @abstractmethod
    def _precompute(self) -> Tuple[torch.Tensor, ...]:
        ...

    def fused_predict_f(self, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        """Predictive mean and (co)variance at Xnew including the mean function; no cache."""
        Xnew = ops.to_device(Xnew)
        mean, cov = self._conditional_fused(Xnew, full_cov=full_cov, full_output_cov=full_output_cov)
        return self._add_mean_function(Xnew, mean), cov

    @abstractmethod
    def _conditional_fused(self, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        ...

    def predict_f(self, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        """Uses the precomputed cache (posteriors.py:285-299)."""
        if self.cache is None:
            raise ValueError(
                "Cache has not been precomputed yet. Call update_cache first or use fused_predict_f")
        Xnew = ops.to_device(Xnew)
        mean, cov = self._conditional_with_precompute(self.cache, Xnew, full_cov=full_cov,
                                                      full_output_cov=full_output_cov)
        return self._add_mean_function(Xnew, mean), cov

    @abstractmethod
    def _conditional_with_precompute(self, cache, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        ...

    def update_cache(self, precompute_cache: Optional[PrecomputeCacheType] = None) -> None:
        """posteriors.py:322-358 (TENSOR and VARIABLE both hold device tensors here)."""
        if precompute_cache is None:
            if self._precompute_cache is None:
                raise ValueError(
                    "You must pass precompute_cache explicitly (the cache had not been updated before).")
            precompute_cache = self._precompute_cache
        else:
            self._precompute_cache = precompute_cache
        if precompute_cache is PrecomputeCacheType.NOCACHE:
            self.cache = None
        elif precompute_cache is PrecomputeCacheType.TENSOR:
            self.cache = tuple(self._precompute())
        elif precompute_cache is PrecomputeCacheType.VARIABLE:
            new = tuple(self._precompute())
            if self.cache is not None and len(self.cache) == len(new) and all(
                    c.shape == n.shape for c, n in zip(self.cache, new)):
                for c, n in zip(self.cache, new):
                    c.copy_(n)  # re-use the existing buffers (tf.Variable.assign in the reference)
            else:
                self.cache = new


class GPRPosterior(AbstractPosterior):
    """posteriors.py:361-443"""

    def __init__(self, kernel, data, likelihood, mean_function, *, precompute_cache) -> None:
        X, Y = data
        super().__init__(kernel, ops.to_device(X), mean_function=mean_function)
        self.Y_data = ops.to_device(Y)
        self.likelihood = likelihood
        self._factor: Optional[Factor] = None
        self._alpha = None
        if precompute_cache is not None:
            self.update_cache(precompute_cache)

    def _err(self):
        return self.Y_data - self.mean_function(self.X_data)

    def _K_plus_noise_into(self, Xsliced, out):
        """K(X, X) + likelihood noise on the diagonal (add_likelihood_noise_cov, utilities/model_utils.py:46-50): the constant rides
        inside the covariance build; a heteroskedastic likelihood adds its per-row variances at the (unsliced) data inputs."""
        if self.likelihood.is_heteroskedastic:
            self.kernel.K_into(Xsliced, None, out, lower_only=True)
            ops.diag_add_(out, self.likelihood.noise_for(self.X_data))
        else:
            self.kernel.K_into(Xsliced, None, out, diag_add=self.likelihood.noise_variance(), lower_only=True)
        return out

    def _precompute(self):
        """cache = (err, Lm) (posteriors.py:415-432); the block inverses and alpha = Lm^-1 err (which rides through
        the factorisation as P extra rows, logdensities.py:150 style) stay alongside Lm."""
        X = self.kernel.slice(self.X_data, None)[0]
        n = X.shape[0]
        err = self._err()
        P = err.shape[1]
        T = torch.empty((n + P, n), dtype=torch.float64, device=X.device)
        self._K_plus_noise_into(X, T[:n])
        T[n:] = err.t()
        invd, info = ops.potrf_(T, n, zero_upper=True)
        ops.check_info(info)
        Lm = T[:n]
        self._factor = Factor(Lm, invd)
        self._alpha = ops.transpose(T[n:])  # [n, P] = Lm^-1 err
        return err, Lm

    def _conditional_with_precompute(self, cache, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        """posteriors.py:384-409"""
        assert_params_false(self._conditional_with_precompute, full_output_cov=full_output_cov)
        err, Lm = cache
        own = self._factor is not None and self._factor.L is Lm
        fac = self._factor if own else Factor(Lm, ops.trtri_blocks(Lm))
        Xf, lead = _flatten_rows(Xnew)
        Xs, Xd = self.kernel.slice(Xf, self.X_data)
        At = self.kernel.K_into(Xs, Xd, None)  # Kmn^T [T, N]
        ops.trsm_(At, fac.L, fac.invd, trans=0)
        alpha = self._alpha if own else None
        return tail_over_batches(
            At, lead, _knn_of(lambda x, fc: self.kernel(x, full_cov=fc), Xf, lead, full_cov),
            lambda A, K: conditional_tail(A, fac, K, err, full_cov=full_cov, q_sqrt=None, white=False, Linv_f=alpha),
            full_cov)

    def _conditional_fused(self, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        """posteriors.py:435-443: Cholesky redone on every call, as in the reference -- here fused
        with the solve of Kmn (extra rows of one trapezoidal factorisation)."""
        assert_params_false(self._conditional_fused, full_output_cov=full_output_cov)
        Xf, lead = _flatten_rows(Xnew)
        Xs, Xd = self.kernel.slice(Xf, self.X_data)
        n, t = Xd.shape[0], Xs.shape[0]
        err = self._err()
        P = err.shape[1]
        # one trapezoid [K + noise I ; Kxs ; err^T]: the factorisation returns A^T = Kxs Lm^-T and alpha^T = (Lm^-1 err)^T
        T = torch.empty((n + t + P, n), dtype=torch.float64, device=Xd.device)
        self._K_plus_noise_into(Xd, T[:n])
        self.kernel.K_into(Xs, Xd, T[n:n + t])
        T[n + t:] = err.t()
        invd, info = ops.potrf_(T, n, zero_upper=True)
        ops.check_info(info)
        fac = Factor(T[:n], invd)
        alpha = ops.transpose(T[n + t:])
        return tail_over_batches(
            T[n:n + t], lead, _knn_of(lambda x, fc: self.kernel(x, full_cov=fc), Xf, lead, full_cov),
            lambda A, K: conditional_tail(A, fac, K, err, full_cov=full_cov, q_sqrt=None, white=False, Linv_f=alpha),
            full_cov)


class BasePosterior(AbstractPosterior):
    """posteriors.py:640-746"""

    def __init__(self, kernel, inducing_variable, q_mu, q_sqrt, whiten: bool = True, mean_function=None,
                 *, precompute_cache):
        super().__init__(kernel, inducing_variable, mean_function=mean_function)
        self.whiten = whiten
        self._set_qdist(q_mu, q_sqrt)
        if precompute_cache is not None:
            self.update_cache(precompute_cache)

    @staticmethod
    def _dev(v):
        from .base import Parameter
        if v is None:
            return None
        if isinstance(v, Parameter):
            return v.device_value()
        return ops.to_device(v)

    def _set_qdist(self, q_mu, q_sqrt) -> None:
        self._q_mu_src, self._q_sqrt_src = q_mu, q_sqrt

    @property
    def q_mu(self) -> torch.Tensor:
        return self._dev(self._q_mu_src)

    @property
    def q_sqrt(self) -> Optional[torch.Tensor]:
        return self._dev(self._q_sqrt_src)

    def _precompute(self):
        """posteriors.py:694-746.  One kernel shared by the latents (Kuu [M,M]): alpha [M,L], Qinv [L,M,M].  Separate
        kernels (Kuu [L,M,M]): the L factorisations are ONE batched launch sequence, then per latent
        alpha_l = solve with q_mu[:, l] -> alpha [L,M,1], Qinv [L,M,M] (the reference's layout, :698-701)."""
        Kuu = covariances.Kuu(self.X_data, self.kernel, jitter=config.default_jitter())
        q_mu, q_sqrt = self.q_mu, self.q_sqrt
        M, Lnum = q_mu.shape
        if Kuu.dim() == 2:
            fac, _ = factor_with_rows(Kuu, None)
            return self._alpha_qinv(fac, q_mu, q_sqrt)
        T = Kuu.contiguous().clone()  # [L, M, M]: batched factorisation (conditionals/util.py:618 is a tf.map_fn loop)
        invd, info = ops.potrf_(T, M, zero_upper=True)
        ops.check_info(info)
        invd = invd.reshape(Lnum, -1)
        alphas, Qinvs = [], []
        for l in range(Lnum):
            qs = None
            if q_sqrt is not None:
                qs = q_sqrt[:, l:l + 1].contiguous() if q_sqrt.dim() == 2 else q_sqrt[l:l + 1]
            a, Q = self._alpha_qinv(Factor(T[l], invd[l]), q_mu[:, l:l + 1].contiguous(), qs)
            alphas.append(a)
            Qinvs.append(Q[0])
        return torch.stack(alphas), torch.stack(Qinvs)  # [L, M, 1], [L, M, M]

    def _alpha_qinv(self, fac: Factor, q_mu: torch.Tensor, q_sqrt: Optional[torch.Tensor]):
        """alpha [M,R] and Qinv [R,M,M] of R latents that share the factor `fac` of Kuu (posteriors.py:703-744)."""
        M, Lnum = q_mu.shape
        LT, invdT = fac.transposed()
        alphaT = ops.transpose(q_mu)  # [L, M] rows
        if not self.whiten:
            ops.trsm_(alphaT, fac.L, fac.invd, trans=0)  # L^-1 q_mu
        ops.trsm_(alphaT, LT, invdT, trans=1)  # L^-T (.)
        alpha = ops.transpose(alphaT)  # [M, L]
        I = torch.eye(M, dtype=torch.float64, device=q_mu.device)
        if q_sqrt is None:
            Bs = I[None]
        else:
            qs = torch.diag_embed(q_sqrt.t().contiguous()) if q_sqrt.dim() == 2 else ops.transpose(
                ops.transpose(q_sqrt, mode=1))  # tril(q_sqrt)
            covs = []
            for l in range(qs.shape[0]):
                G = qs[l].contiguous()
                if not self.whiten:
                    Gt = ops.transpose(G)  # rows = q_sqrt^T
                    ops.trsm_(Gt, fac.L, fac.invd, trans=0)  # (L^-1 q_sqrt)^T
                    G = ops.transpose(Gt)
                covs.append(ops.gemm_nt(G, G))  # (L^-1) S (L^-T)  or  S
            Bs = I[None] - torch.stack(covs)
        Qinv = []
        for b in Bs:
            Y = b.contiguous().clone()
            ops.trsm_(Y, LT, invdT, trans=1)  # B L^-1
            Yt = ops.transpose(Y)
            ops.trsm_(Yt, LT, invdT, trans=1)  # (L^-T B L^-1)^T, symmetric
            Qinv.append(Yt)
        Qinv = torch.stack(Qinv)
        if Qinv.shape[0] != Lnum:
            Qinv = Qinv.expand(Lnum, M, M).contiguous()
        return alpha, Qinv


class IndependentPosterior(BasePosterior):
    """posteriors.py:749-822"""

    def _post_process_mean_and_cov(self, mean, cov, full_cov: bool, full_output_cov: bool):
        return mean, expand_independent_outputs(cov, full_cov, full_output_cov)

    def _get_Kff(self, Xnew, full_cov: bool):
        if isinstance(self.kernel, SeparateIndependent):
            return torch.stack([k(Xnew, full_cov=full_cov) for k in self.kernel.kernels], dim=0)
        elif isinstance(self.kernel, MultioutputKernel):
            return self.kernel.kernel(Xnew, full_cov=full_cov)
        return self.kernel(Xnew, full_cov=full_cov)

    def _conditional_with_precompute(self, cache, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        """posteriors.py:794-822 in the row-major form: mean = Kfu alpha, cov = Kff - rowdot(Kfu Qinv, Kfu).  One kernel:
        Kfu [N,M], alpha [M,L]; separate kernels: Kfu [L,N,M], alpha [L,M,1] (:813-815).  Xnew may carry leading batch
        dims; with full_cov the [T,T] blocks are formed per batch element."""
        alpha, Qinv = cache
        Xf, lead = _flatten_rows(Xnew)
        Kfu = covariances.Kfu(self.X_data, self.kernel, Xf)
        separate = Kfu.dim() == 3
        # Kff carries a latent axis only for separate KERNELS; a shared kernel over separate inducing variables has
        # Kfu [L,N,M] but one Kff, broadcast over the latents (posteriors.py:794-822)
        kff_per_latent = isinstance(self.kernel, SeparateIndependent)
        Lnum = Qinv.shape[0]
        if separate:
            mean = torch.stack([ops.row_stats(Kfu[l], V=alpha[l].contiguous(), want_sumsq=False)[1][:, 0] for l in range(Lnum)],
                               dim=-1)
        else:
            _, mean, _ = ops.row_stats(Kfu, V=alpha.contiguous(), want_sumsq=False)
        Ws = [ops.gemm_nt(Kfu[l] if separate else Kfu, Qinv[l]) for l in range(Lnum)]  # Kfu Qinv (Qinv symmetric)
        if not full_cov:
            Kff = self._get_Kff(Xf, False)  # [N] or [L, N]
            cov = torch.stack([(Kff[l] if kff_per_latent else Kff) - ops.row_dot(Ws[l], Kfu[l] if separate else Kfu)
                               for l in range(Lnum)], dim=-1)
            if len(lead) > 1:
                mean, cov = mean.reshape(*lead, -1), cov.reshape(*lead, -1)
            return self._post_process_mean_and_cov(mean, cov, full_cov, full_output_cov)
        T = lead[-1]
        nb = Xf.shape[0] // T
        blocks = []
        for b in range(nb):
            r = slice(b * T, (b + 1) * T)
            Kff = self._get_Kff(Xf[r], True)  # [T, T] or [L, T, T]
            blocks.append(torch.stack([(Kff[l] if kff_per_latent else Kff)
                                       - ops.gemm_nt(Ws[l][r], (Kfu[l] if separate else Kfu)[r]) for l in range(Lnum)], dim=0))
        cov = torch.stack(blocks)  # [nb, L, T, T]
        if len(lead) > 1:
            mean, cov = mean.reshape(*lead, -1), cov.reshape(*lead[:-1], Lnum, T, T)
        else:
            cov = cov[0]
        return self._post_process_mean_and_cov(mean, cov, full_cov, full_output_cov)

    # shared machinery of the fused paths ---------------------------------------------------------
    def _fused_single_kernel(self, kernel: Kernel, Z: torch.Tensor, Xnew, full_cov: bool):
        """Kuu (+jitter), Kuf and base_conditional for ONE kernel shared by all latents
        (posteriors.py:828-841 / 849-861), as one trapezoidal factorisation built in place."""
        Xf, lead = _flatten_rows(Xnew)
        Xs, Zs = kernel.slice(Xf, Z)
        M, N = Zs.shape[0], Xs.shape[0]
        T = torch.empty((M + N, M), dtype=torch.float64, device=Zs.device)
        kernel.K_into(Zs, None, T[:M], diag_add=config.default_jitter(), lower_only=True)
        kernel.K_into(Xs, Zs, T[M:])
        invd, info = ops.potrf_(T, M, zero_upper=True)
        ops.check_info(info)
        fac, q_mu, q_sqrt = Factor(T[:M], invd), self.q_mu, self.q_sqrt
        return tail_over_batches(
            T[M:], lead, _knn_of(lambda x, fc: kernel(x, full_cov=fc), Xf, lead, full_cov),
            lambda A, K: conditional_tail(A, fac, K, q_mu, full_cov=full_cov, q_sqrt=q_sqrt, white=self.whiten), full_cov)


class IndependentPosteriorSingleOutput(IndependentPosterior):
    """posteriors.py:825-841"""

    def _conditional_fused(self, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        fmean, fvar = self._fused_single_kernel(self.kernel, self.X_data.Z.device_value(), Xnew, full_cov)
        return self._post_process_mean_and_cov(fmean, fvar, full_cov, full_output_cov)


class IndependentPosteriorMultiOutput(IndependentPosterior):
    """posteriors.py:844-887"""

    def _conditional_fused(self, Xnew, full_cov: bool = False, full_output_cov: bool = False):
        if isinstance(self.X_data, SharedIndependentInducingVariables) and isinstance(self.kernel, SharedIndependent):
            fmean, fvar = self._fused_single_kernel(self.kernel.kernel,
                                                    self.X_data.inducing_variable.Z.device_value(), Xnew,
                                                    full_cov)
        else:
            Xf, lead = _flatten_rows(Xnew)
            pairs = covariances._pairs(self.X_data, self.kernel)  # (Z_p, kernel_p) per latent
            kernel_list = [k for _, k in pairs]
            q_mu, q_sqrt = self.q_mu, self.q_sqrt

            def one(Xr):
                # Kuu_p + jitter I and Kfu_p built STRAIGHT into the batched trapezoid [P, M + N, M] (posteriors.py:862-887
                # stacks [P,M,M] and [P,M,N] tensors first): one batched factorisation + solve, no intermediate copies
                M, N = pairs[0][0].shape[0], Xr.shape[0]
                T = torch.empty((len(pairs), M + N, M), dtype=torch.float64, device=Xr.device)
                for p, (z, k) in enumerate(pairs):
                    Xs, Zs = k.slice(Xr, z)
                    k.K_into(Zs, None, T[p, :M], diag_add=config.default_jitter(), lower_only=True)
                    k.K_into(Xs, Zs, T[p, M:])
                Knns = torch.stack([k(Xr, full_cov=full_cov) for k in kernel_list], dim=0)
                return separate_independent_trapezoid_tail(T, M, Knns, q_mu, full_cov=full_cov, q_sqrt=q_sqrt,
                                                           white=self.whiten)

            if len(lead) == 1:
                fmean, fvar = one(Xf)
            elif not full_cov:
                fmean, fvar = one(Xf)
                fmean, fvar = fmean.reshape(*lead, -1), fvar.reshape(*lead, -1)
            else:  # [batch..., P, T, T]: one block per batch element
                T = lead[-1]
                res = [one(Xf[b * T:(b + 1) * T].contiguous()) for b in range(Xf.shape[0] // T)]
                fmean = torch.stack([r[0] for r in res]).reshape(*lead, -1)
                fvar = torch.stack([r[1] for r in res])
                fvar = fvar.reshape(*lead[:-1], *fvar.shape[1:])
        return self._post_process_mean_and_cov(fmean, fvar, full_cov, full_output_cov)


def get_posterior_class(kernel, inducing_variable) -> Type[BasePosterior]:
    """posteriors.py:1039-1086 (rows on the path; everything else is out of scope)."""
    if isinstance(kernel, (SharedIndependent, SeparateIndependent)):
        if isinstance(inducing_variable, (SeparateIndependentInducingVariables,
                                          SharedIndependentInducingVariables)):
            return IndependentPosteriorMultiOutput
        raise NotImplementedError(
            "multi-output kernels with plain InducingPoints (FullyCorrelatedPosterior) are out of scope")
    if isinstance(kernel, MultioutputKernel):
        raise NotImplementedError(f"posterior for {type(kernel).__name__} is out of scope")
    if isinstance(kernel, Kernel) and isinstance(inducing_variable, InducingVariables):
        return IndependentPosteriorSingleOutput
    raise NotImplementedError(
        f"no posterior registered for ({type(kernel).__name__}, {type(inducing_variable).__name__})")


def create_posterior(kernel, inducing_variable, q_mu, q_sqrt, whiten, mean_function=None,
                     precompute_cache: Union[PrecomputeCacheType, str, None] = PrecomputeCacheType.TENSOR):
    """posteriors.py:1089-1108"""
    posterior_class = get_posterior_class(kernel, inducing_variable)
    precompute_cache = _validate_precompute_cache_type(precompute_cache)
    return posterior_class(kernel, inducing_variable, q_mu, q_sqrt, whiten, mean_function,
                           precompute_cache=precompute_cache)
