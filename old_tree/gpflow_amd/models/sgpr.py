"""SGPR -- sparse GP regression, Titsias 2009 (gpflow/models/sgpr.py:36-384), SURVEY 8f row 3: the same three device
primitives as the SVGP step (covariance builder, trapezoidal Cholesky with fused solve, fp64 MFMA GEMM), but the data
axis is the WHOLE data set and the exchange between row shards is a matrix, not a scalar.

With the rows of (X, Y) sharded over ranks, a rank computes from its shard

    At = Kfu Lm^-T [n, M]        (Lm = chol(Kuu + jitter I), replicated;  the reference's A is At^T / sigma)
    S  = At^T At [M, M],   a = At^T (Y - m) [M, P],   e2 = |Y - m|^2,   q = |At|_F^2

and ONE all-reduce of the packed (S lower tiles as a full block, a, e2, q) -- M^2 + M P + 2 doubles, 33.6 MB at M = 2048
-- gives every rank the statistics of the full data set; the M x M tail (B = I + S / s2, its Cholesky with the fused
solve for c) is replicated.  Single-process use needs no collective.  Constant noise variance, stationary kernel,
constant / zero mean (anything else raises NotImplementedError: not on the path of this row).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from .. import config, gradients, ops
from ..inducing_variables import InducingPoints, inducingpoint_wrapper
from ..kernels import Kernel
from ..kernels.stationaries import Stationary
from ..likelihoods import Gaussian
from ..mean_functions import MeanFunction
from ..posteriors import assert_params_false
from .model import GPModel
from .training_mixins import InternalDataTrainingLossMixin

LOG2PI = float(np.log(2.0 * np.pi))


def shard_statistics(Z: torch.Tensor, X: torch.Tensor, Y: torch.Tensor, *, spec: "gradients.KernelSpec" = None, jitter: float,
                     mean_const: float, variance: float = None, lengthscales=None, family: str = "SquaredExponential",
                     noise_rows: Optional[torch.Tensor] = None):
    """(L [M,M] lower with zero upper, invd, packed statistics [M*M + M*P + 2]) of one row shard (see module doc).
    spec: the covariance function (one stationary kernel, or a Sum / Product of them -- gradients.KernelSpec).
    noise_rows [n]: one noise variance per data row (a heteroskedastic Gaussian likelihood, sgpr.py:207-211: A = L^-1 Kuf / sigma,
    err / sigma) -- the rows of At and err are scaled by 1 / sigma_n, so every statistic is the reference's with sigma^2 = 1, and two
    more scalars ride behind them: sum_n log sigma_n^2 and sum_n 1 / sigma_n^2 (the trace_k and log_sigma_sq terms, :236-247), and a
    third for the upper bound: |At|_F^2 of the UNSCALED rows (its c = sum Kdiag - sum A^2, :126)."""
    M, n, P = Z.shape[0], X.shape[0], Y.shape[1]
    if spec is None:
        spec = gradients.KernelSpec.single(variance, lengthscales, family)
    T = torch.empty((M + n, M), dtype=torch.float64, device=Z.device)
    spec.build(Z, None, T[:M], diag_add=jitter)                                         # Kuu + jitter I  (sgpr.py:200)
    if n:
        spec.build(X, Z, T[M:])                                                         # Kfu             (:199)
    invd, info = ops.potrf_(T, M, zero_upper=True)                                      # L (:201); At = Kfu L^-T (:204)
    ops.check_info(info)
    L, At = T[:M], T[M:]
    het = noise_rows is not None
    packed = torch.zeros(M * M + M * P + (5 if het else 2), dtype=torch.float64, device=Z.device)
    if n:
        err = (Y - mean_const).contiguous()
        if het:
            packed[M * M + M * P + 4] = ops.sumsq(At)[0]
            w = 1.0 / noise_rows.reshape(-1)
            sw = torch.sqrt(w)
            At = At * sw[:, None]                                                       # rows of A^T / sigma_n (elementwise glue)
            err = err * sw[:, None]
        A = ops.transpose(At)                                                           # [M, n]
        S = gradients.splitk_gemm_nt(A, A, c_lower=True)                                # At^T At, lower tiles (:205)
        packed[:M * M] = torch.tril(S).reshape(-1)
        packed[M * M:M * M + M * P] = gradients.splitk_gemm_nt(A, err.t().contiguous()).reshape(-1)   # At^T err (:268)
        o = M * M + M * P
        packed[o] = ops.sumsq(err)[0]
        packed[o + 1] = ops.sumsq(At)[0]
        if het:
            packed[o + 2] = -torch.log(w).sum()
            packed[o + 3] = w.sum()
    return L, invd, packed


def tail_factor(packed: torch.Tensor, M: int, P: int, scale: float):
    """LB = chol(I + S * scale) with the fused solve  c^T = (a * scale)^T LB^-T  (sgpr.py:206-207, 268-269 with
    scale = 1 / s2; the upper bound reuses it with 1 / cn_var).  Returns (LB, invdB, c^T [P, M])."""
    S = packed[:M * M].reshape(M, M)
    T2 = torch.empty((M + P, M), dtype=torch.float64, device=packed.device)
    T2[:M] = S * scale
    T2[:M].diagonal().add_(1.0)                                                         # add_noise_cov(AAT, 1)
    T2[M:] = packed[M * M:M * M + M * P].reshape(M, P).t() * scale
    invdB, info = ops.potrf_(T2, M, zero_upper=True)
    ops.check_info(info)
    return T2[:M], invdB, T2[M:]


def elbo_from_statistics(packed: torch.Tensor, M: int, P: int, N: int, *, variance: float, noise_variance):
    """sgpr.py:214-290 from the (all-reduced) statistics; the reference's A carries 1/sigma, here it is explicit.
    noise_variance None: heteroskedastic statistics (rows already scaled by 1 / sigma_n; shard_statistics(noise_rows=...))."""
    o = M * M + M * P
    if noise_variance is None:
        LB, _, ct = tail_factor(packed, M, P, 1.0)
        half_logdet_b = ops.sum_log_diag(LB)[0]
        trace = variance * packed[o + 3] - packed[o + 1]                                # sum kdiag / sigma^2 - tr(A A^T)   :236-242
        logdet = -P * (half_logdet_b + 0.5 * packed[o + 2] + 0.5 * trace)               # :248-251 with sum log sigma_n^2
        quad = -0.5 * (packed[o] - ops.sumsq(ct)[0])
        return -0.5 * N * P * LOG2PI + logdet + quad
    s2 = noise_variance
    LB, _, ct = tail_factor(packed, M, P, 1.0 / s2)
    half_logdet_b = ops.sum_log_diag(LB)[0]                                             # :245
    trace = N * variance / s2 - packed[o + 1] / s2                                      # :236-242
    logdet = -P * (half_logdet_b + 0.5 * N * float(np.log(s2)) + 0.5 * trace)           # :248-251
    # the reference's c = LB^-1 A err with A = At^T / sigma and err / sigma: both sigma factors sit in a / s2, so ct IS c^T
    quad = -0.5 * (packed[o] / s2 - ops.sumsq(ct)[0])                                    # :272-276
    return -0.5 * N * P * LOG2PI + logdet + quad                                        # :287-290


def upper_bound_from_statistics(packed: torch.Tensor, M: int, N: int, *, variance: float, noise_variance: float):
    """sgpr.py:85-148 (single-output form, as written in the reference)."""
    s2 = noise_variance
    LB, _, _ = tail_factor(packed, M, 1, 1.0 / s2)
    c_tr = N * variance - packed[M * M + M + 1]                                         # :121
    cn_var = float(s2 + c_tr)                                                           # :124 (host scalar: one read-back)
    _, _, vt = tail_factor(packed, M, 1, 1.0 / cn_var)                                  # LC, v (:130-137)
    const = -0.5 * N * float(np.log(2 * np.pi * s2))
    logdet = -ops.sum_log_diag(LB)[0]
    quad = -0.5 * packed[M * M + M] / cn_var + 0.5 * ops.sumsq(vt)[0]
    return const + logdet + quad


def upper_bound_heteroskedastic(packed: torch.Tensor, packed_cn: torch.Tensor, M: int, N: int):
    """sgpr.py:85-148 with one sigma_n^2 per data row.  `packed`: the statistics of shard_statistics(noise_rows = sigma_n^2) -- LB and
    sum log sigma_n^2 come from them; `packed_cn`: a SECOND pass over the rows with noise_rows = sigma_n^2 + c (:129-131: A_cn, err / cn_std),
    whose rows are scaled by 1 / cn_std, so LC, v and the quadratic term are the constant-noise ones with variance 1."""
    o = M * M + M
    LB, _, _ = tail_factor(packed, M, 1, 1.0)                                           # :121-123 (rows already carry 1 / sigma_n)
    const = -0.5 * N * LOG2PI - 0.5 * packed[o + 2]                                     # :132  -0.5 sum log(2 pi sigma_n^2)
    logdet = -ops.sum_log_diag(LB)[0]                                                   # :133
    _, _, vt = tail_factor(packed_cn, M, 1, 1.0)                                        # LC, v (:139-142)
    quad = -0.5 * packed_cn[o] + 0.5 * ops.sumsq(vt)[0]                                 # :143-145
    return const + logdet + quad


class SGPR(GPModel, InternalDataTrainingLossMixin):
    def __init__(self, data, kernel: Kernel, inducing_variable, *, mean_function: Optional[MeanFunction] = None,
                 num_latent_gps: Optional[int] = None, noise_variance=None, likelihood: Optional[Gaussian] = None,
                 sharded: bool = False, group=None):
        """sgpr.py:46-81.  `sharded=True` (NEW: the reference is single process): `data` is THIS rank's row shard and
        the sufficient statistics are summed over the ranks of `group` (torch.distributed; RCCL on the GPUs)."""
        assert (noise_variance is None) or (likelihood is None), "Cannot set both `noise_variance` and `likelihood`."
        if likelihood is None:
            likelihood = Gaussian(1.0 if noise_variance is None else noise_variance)
        X, Y = data
        self.data = (ops.to_device(X), ops.to_device(Y))
        if self.data[0].dim() != 2 or self.data[1].dim() != 2 or self.data[0].shape[0] != self.data[1].shape[0]:
            raise ValueError("data must be (X [N,D], Y [N,P])")
        P = self.data[1].shape[-1] if num_latent_gps is None else num_latent_gps
        super().__init__(kernel, likelihood, mean_function, num_latent_gps=P)
        self.inducing_variable = inducingpoint_wrapper(inducing_variable)
        self.sharded, self.group = bool(sharded), group
        self.num_data = self._global_rows()

    # ---- plumbing --------------------------------------------------------------------------------
    def _global_rows(self) -> int:
        n = int(self.data[0].shape[0])
        if self.sharded:
            import torch.distributed as dist
            t = torch.tensor([float(n)], dtype=torch.float64, device=self.data[0].device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            n = int(round(float(t.cpu()[0])))
        return n

    def _config(self):
        k, iv, lik = self.kernel, self.inducing_variable, self.likelihood
        c = self.mean_function.constant_value()
        from ..kernels.base import Combination, gradient_spec
        if not (isinstance(k, (Stationary, Combination)) and isinstance(iv, InducingPoints) and isinstance(lik, Gaussian)
                and c is not None):
            raise NotImplementedError("SGPR here: stationary kernel (or a Sum / Product of them), InducingPoints, Gaussian "
                                      "likelihood, constant mean")
        # s2: the constant noise variance, or None for a heteroskedastic likelihood (variance / scale a Function of the inputs,
        # likelihoods/scalar_continuous.py:52-111): then `self._noise_rows()` is sigma_n^2 at the data inputs (sgpr.py:207)
        s2 = None if lik.is_heteroskedastic else lik.noise_variance()
        if isinstance(k, Combination):
            # members slice for themselves (kernels/base.py:283-293): the spec works on the full columns
            spec, _ = gradient_spec(k, self.data[0].shape[1])
            return spec, self.data[0].contiguous(), iv.Z.device_value().contiguous(), float(c), s2
        family, var, ls = k.hyper()
        X, Z = k.slice(self.data[0], iv.Z.device_value())
        return gradients.KernelSpec.single(var, ls, family), X, Z, float(c), s2

    def _noise_rows(self):
        return self.likelihood.noise_for(self.data[0]) if self.likelihood.is_heteroskedastic else None

    def _slice_new(self, Xn):
        """new inputs as the covariance spec expects them: sliced by a single kernel's active_dims, untouched for a combination"""
        from ..kernels.base import Combination
        return Xn.contiguous() if isinstance(self.kernel, Combination) else self.kernel.slice(Xn, None)[0]

    def _statistics(self):
        kw, X, Z, c, s2 = self._config()
        L, invd, packed = shard_statistics(Z, X, self.data[1], jitter=config.default_jitter(), mean_const=c, spec=kw,
                                           noise_rows=self._noise_rows())
        if self.sharded:
            import torch.distributed as dist
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
        return kw, Z, c, s2, L, invd, packed

    # ---- objectives ------------------------------------------------------------------------------
    def maximum_log_likelihood_objective(self):
        return self.elbo()

    def elbo(self) -> torch.Tensor:
        """sgpr.py:279-290"""
        kw, Z, c, s2, L, invd, packed = self._statistics()
        return elbo_from_statistics(packed, Z.shape[0], self.data[1].shape[1], self.num_data, variance=kw.kdiag(),
                                    noise_variance=s2)

    def upper_bound(self) -> torch.Tensor:
        """sgpr.py:85-148"""
        if self.data[1].shape[1] != 1:
            raise NotImplementedError("upper_bound is written for a single output column in the reference (sgpr.py:126)")
        kw, Z, c, s2, L, invd, packed = self._statistics()
        M = Z.shape[0]
        if s2 is None:
            # sgpr.py:124-131 rescales every row by ITS sigma_n^2 + c, where c needs the statistics of ALL rows: a second pass
            # (covariances, solve, statistics again -- the same cost as the first; one more all-reduce on a sharded model)
            c_tr = self.num_data * kw.kdiag() - float(packed[M * M + M + 4].cpu())      # :126 (host scalar: one read-back)
            _, X, _, _, _ = self._config()
            _, _, packed_cn = shard_statistics(Z, X, self.data[1], jitter=config.default_jitter(), mean_const=c, spec=kw,
                                               noise_rows=self._noise_rows() + c_tr)
            if self.sharded:
                import torch.distributed as dist
                dist.all_reduce(packed_cn, op=dist.ReduceOp.SUM, group=self.group)
            return upper_bound_heteroskedastic(packed, packed_cn, M, self.num_data)
        return upper_bound_from_statistics(packed, M, self.num_data, variance=kw.kdiag(), noise_variance=s2)

    def objective_and_grad(self):
        """(ELBO as a float, {Parameter: dELBO/d(unconstrained value)}) for the trainable parameters among kernel variance,
        lengthscales, noise variance, Z and a Constant mean -- the gradient `optimizers/scipy.py:322-331` takes from TF
        (gradients.sgpr_elbo_and_grad; SquaredExponential or Matern12 / 32 / 52 kernel, `active_dims` allowed: dELBO/dZ
        is zero in the columns the kernel does not see).  On a row-sharded model every rank gets the complete ELBO and
        gradient: two all-reduces of M^2 + O(M) doubles per evaluation (gradients.sgpr_elbo_and_grad)."""
        from ..kernels.stationaries import IsotropicStationary
        from ..mean_functions import Constant
        from .svgp import SVGP
        from ..kernels.base import Combination, gradient_spec
        kw, Xc, Zc, c, s2 = self._config()
        het = s2 is None
        if not het and not self.likelihood.has_variance_parameter:
            raise NotImplementedError("gradients: the reverse pass takes a noise variance held as a `variance` Parameter, or a noise "
                                      "Function of the inputs")
        if het:   # one sigma_n^2 per row of this (shard of the) data; dF/d sigma_n^2 comes back per row (gradients.sgpr_elbo_and_grad)
            s2 = self._noise_rows().reshape(-1).contiguous()

        def noise_pairs(g_noise):
            if not het:
                return [(self.likelihood.variance, g_noise.cpu().numpy())]
            out_ = []
            for par, gv in self.likelihood.noise_param_grads(self.data[0], g_noise):   # chain rule through the noise function: this shard's rows
                gv = gv.contiguous()
                if self.sharded:
                    import torch.distributed as dist
                    dist.all_reduce(gv, op=dist.ReduceOp.SUM, group=self.group)
                out_.append((par, gv.cpu().numpy()))
            return out_
        if isinstance(self.kernel, Combination):
            # a Sum / Product of stationary kernels (members possibly over different active_dims): the members' adjoints one by one
            spec, members = gradient_spec(self.kernel, self.data[0].shape[1])
            F, g, info = gradients.sgpr_elbo_and_grad(Zc, Xc, self.data[1], noise_variance=s2, jitter=config.default_jitter(),
                                                      mean_const=c, sharded=self.sharded, group=self.group,
                                                      num_data=self.num_data, kernel_spec=spec)
            ops.check_info(info)
            gv = g["variance"].cpu().numpy()
            host = {n: t.cpu().numpy() for n, t in g.items() if n not in ("variance", "lengthscales", "noise_variance")}
            pairs = []
            for i, (pv, pl) in enumerate(members):
                pairs += [(pv, gv[i]), (pl, g["lengthscales"][i].cpu().numpy())]
            pairs += noise_pairs(g["noise_variance"]) + [(self.inducing_variable.Z, host["Z"])]
        else:
            if not (isinstance(self.kernel, IsotropicStationary) and self.kernel.family in ops.KERNEL_FAMILIES):
                raise NotImplementedError("gradients: SquaredExponential / Matern kernel")
            family, var, ls = self.kernel.hyper()
            Z, X, scatter = SVGP._sliced(self.kernel, self.inducing_variable.Z.device_value(), self.data[0])
            F, g, info = gradients.sgpr_elbo_and_grad(Z, X, self.data[1], noise_variance=s2, jitter=config.default_jitter(),
                                                      mean_const=c, sharded=self.sharded, group=self.group,
                                                      num_data=self.num_data, variance=var, lengthscales=ls, family=family)
            ops.check_info(info)
            host = {n: (scatter(t) if n == "Z" else t).cpu().numpy() for n, t in g.items() if n != "noise_variance"}
            pairs = [(self.kernel.variance, host["variance"]), (self.kernel.lengthscales, host["lengthscales"]),
                     (self.inducing_variable.Z, host["Z"])] + noise_pairs(g["noise_variance"])
        if isinstance(self.mean_function, Constant) and hasattr(self.mean_function, "c"):
            pairs.append((self.mean_function.c, host["mean_const"]))
        out = {}
        for par, gc in pairs:
            if par.trainable:
                u = par.unconstrained_variable
                gu = np.asarray(gc, dtype=np.float64).reshape(u.shape) * par.transform.forward_grad(u)
                out[par] = out[par] + gu if par in out else gu
        return self._add_log_prior(float(F.cpu()[0]), out)   # (+ log prior density: -training_loss, model.py:56-76)

    # ---- prediction ------------------------------------------------------------------------------
    def predict_f(self, Xnew, full_cov: bool = False, full_output_cov: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """sgpr.py:292-345"""
        assert_params_false(self.predict_f, full_output_cov=full_output_cov)
        kw, Z, c, s2, L, invd, packed = self._statistics()
        M, P = Z.shape[0], self.data[1].shape[1]
        LB, invdB, ct = tail_factor(packed, M, P, 1.0 if s2 is None else 1.0 / s2)   # (heteroskedastic: rows already scaled)
        # the reference's c carries one factor sigma more than ct (A = At^T / sigma, err / sigma): mean = tmp2^T c with
        # tmp2 free of sigma, so mean = tmp2^T ct  exactly as below
        Xn = ops.to_device(Xnew)
        lead = Xn.shape[:-1]
        Xn2 = self._slice_new(Xn.reshape(-1, Xn.shape[-1]))
        t1 = kw.build(Xn2, Z)                                                           # Kus^T [T, M]
        ops.trsm_(t1, L, invd, trans=0)                                                 # tmp1^T = Kus^T L^-T
        t2 = t1.clone()
        ops.trsm_(t2, LB, invdB, trans=0)                                               # tmp2^T
        mean = ops.gemm_nt(t2, ct.contiguous()) + c                                     # [T, P]
        if full_cov:
            var = kw.build(Xn2, None)
            ops.gemm_nt(t2, t2, alpha=1.0, beta=1.0, C=var)
            ops.gemm_nt(t1, t1, alpha=-1.0, beta=1.0, C=var)
            var = var[None].expand(self.num_latent_gps, -1, -1).contiguous()
            return mean.reshape(lead + (P,)), var
        v = kw.kdiag() + ops.row_stats(t2)[0] - ops.row_stats(t1)[0]
        var = v[:, None].expand(-1, self.num_latent_gps).contiguous()
        return mean.reshape(lead + (P,)), var.reshape(lead + (self.num_latent_gps,))

    def compute_qu(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """sgpr.py:351-384: mean [M, P] and covariance [M, M] of q(u) (single process)."""
        if self.sharded:
            raise NotImplementedError("compute_qu on a sharded model")
        kw, X, Z, c, s2 = self._config()
        M, P = Z.shape[0], self.data[1].shape[1]
        Kfu = kw.build(X, Z)
        err = (self.data[1] - c).contiguous()
        if s2 is None:   # scaled_kuf = kuf / std, scaled_err = err / std per data row (sgpr.py:366-377)
            sw = torch.rsqrt(self._noise_rows())
            Kfu = Kfu * sw[:, None]
            err = err * sw[:, None]
            s2 = 1.0
        Kuf = ops.transpose(Kfu)                                                        # [M, N]
        T = torch.empty((M + M + P, M), dtype=torch.float64, device=Z.device)
        kuu = kw.build(Z, None, diag_add=config.default_jitter())
        T[:M] = kuu + gradients.splitk_gemm_nt(Kuf, Kuf, c_lower=True) / s2   # sig (lower triangle is read)
        T[M:2 * M] = kuu                                                                # rows -> kuu sig_sqrt^-T
        T[2 * M:] = gradients.splitk_gemm_nt(Kuf, err.t().contiguous()).t() / s2         # (scaled_kuf scaled_err)^T
        _, info = ops.potrf_(T, M, zero_upper=True)
        ops.check_info(info)
        Sm, vt = T[M:2 * M].contiguous(), T[2 * M:].contiguous()                        # sig_sqrt_kuu^T, v^T
        return ops.gemm_nt(Sm, vt), ops.gemm_nt(Sm, Sm)
