"""Device-resident training loop for the SVGP, whitened or not (SquaredExponential kernel, Gaussian likelihood) -- the caller
of the hot path that SURVEY 8f row 1 names: the Adam loop of `gps_for_big_data.pct.py:207-228`
(`tf.optimizers.Adam().minimize(model.training_loss_closure(iter), model.trainable_variables)`).

The reference keeps variables in TF and differentiates through them; here the big variables (q_mu, q_sqrt, Z) and their
Adam moments stay in HBM, the handful of scalar hyper-parameters live on the host in unconstrained form (their
constrained values are C-ABI arguments), and one step is
    gradients.svgp_elbo_and_grad  ->  (multi-GPU: one all-reduce of the packed gradient)  ->  Adam update.
The update rule and defaults are tf.keras Adam's (lr 1e-3, beta 0.9 / 0.999, epsilon 1e-7, bias-corrected step size).
The Adam arithmetic on the device tensors is one launch per variable (`gpk_adam_step`).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import config, distributed, gradients, ops


class _Adam:
    def __init__(self, lr: float, b1: float, b2: float, eps: float):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m: Dict[str, object] = {}
        self.v: Dict[str, object] = {}

    def step_size(self) -> float:
        return self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)

    def update_device(self, name: str, p: torch.Tensor, g: torch.Tensor) -> None:
        if name not in self.m:
            self.m[name], self.v[name] = torch.zeros_like(p), torch.zeros_like(p)
        m, v = self.m[name], self.v[name]
        # (one launch per variable -- gpk_adam_step -- instead of seven elementwise passes over q_sqrt's M^2 entries)
        ops.adam_step_(p, g.contiguous(), m, v, beta1=self.b1, beta2=self.b2, epsilon=self.eps, step=self.step_size())

    def update_host(self, name: str, p: np.ndarray, g: np.ndarray) -> np.ndarray:
        if name not in self.m:
            self.m[name], self.v[name] = np.zeros_like(p), np.zeros_like(p)
        self.m[name] = self.b1 * self.m[name] + (1.0 - self.b1) * g
        self.v[name] = self.b2 * self.v[name] + (1.0 - self.b2) * g * g
        return p - self.step_size() * self.m[name] / (np.sqrt(self.v[name]) + self.eps)


class SVGPTrainer:
    """Adam on -ELBO for `model` (SVGP: whitened or un-whitened, full or diagonal q_sqrt, SquaredExponential / Matern kernel with
    or without `active_dims` -- or a Sum / Product of such kernels, flat or nested, or one kernel per latent over shared inducing points --, Gaussian likelihood with a constant variance or a noise Function of the inputs, InducingPoints, constant or zero mean).  Honours `Parameter.trainable` (including a trainable
    `Constant.c`).  A step whose Kuu factorisation fails raises and leaves variables and Adam moments untouched.

        trainer = SVGPTrainer(model, learning_rate=1e-3)
        for Xb, Yb in batches:            # device or host arrays; with torch.distributed initialised each rank passes
            elbo = trainer.step((Xb, Yb))  # ITS row shard of the global minibatch (global_batch = total rows)
        trainer.sync_to_model()           # write the trained values back into the model's Parameters
    """

    def __init__(self, model, *, learning_rate: float = 1e-3, beta_1: float = 0.9, beta_2: float = 0.999,
                 epsilon: float = 1e-7, natgrad_gamma: Optional[float] = None, group=None):
        """natgrad_gamma: if given, (q_mu, q_sqrt) take a natural-gradient step of that size per iteration
        (optimizers/natgrad.py; natgrad.natgrad_update on the device) and Adam handles the remaining parameters -- the
        hybrid recipe of the reference's natural-gradient notebook, from ONE gradient evaluation per step."""
        # NotImplementedError outside the scope of the reverse pass
        # a Sum / Product of stationary kernels (flat or nested, members over their own columns): the covariance spec is rebuilt from the
        # trainer's current member values every step (gradients.KernelSpec); one stationary kernel keeps the scalar arguments
        from .kernels.base import Combination, gradient_spec
        self.combo = None
        self.sep = None
        from .kernels import SeparateIndependent
        if isinstance(model.kernel, SeparateIndependent):
            # one kernel per latent over SHARED inducing points (BASELINE config C5, separate): P single-output problems that share Z, the
            # likelihood, the mean constant and the minibatch rows; their objectives and shared gradients add up (conditionals/util.py:566-629)
            from .inducing_variables import SharedIndependentInducingVariables
            if not isinstance(model.inducing_variable, SharedIndependentInducingVariables):
                raise NotImplementedError("the trainer with separate kernels per latent: shared inducing points")
            sepc = model._separate_gradient_config()             # (raises outside the reverse pass)
            members, c = sepc
            self.sep = [k_ for k_, _ in members]
            iv = members[0][1]
            k = None
        elif isinstance(model.kernel, Combination):
            from .inducing_variables import InducingPoints
            from .likelihoods import Gaussian
            iv, c = model.inducing_variable, model.mean_function.constant_value()
            lik0 = model.likelihood
            if not (isinstance(iv, InducingPoints) and c is not None and isinstance(lik0, Gaussian)
                    and (lik0.has_variance_parameter or lik0.is_heteroskedastic)):
                raise NotImplementedError("the trainer with a kernel combination: Gaussian likelihood, InducingPoints, constant mean")
            spec0, members = gradient_spec(model.kernel, int(iv.Z.shape[1]))   # (raises for members outside the reverse pass)
            self.combo = (spec0, members)
            k = None
        else:
            k, iv, c = model.gradient_config(allow_active_dims=True, allow_q_diag=True, allow_heteroskedastic=True)
        lik = model.likelihood
        self.kernel = k
        self.q_diag = model.q_sqrt.numpy().ndim == 2
        if self.q_diag and natgrad_gamma is not None:
            raise NotImplementedError("natural gradients need the full q_sqrt [P, M, M] (optimizers/natgrad.py:280-368)")
        # priors (MAP, model.py:47-76: the loss is -(ELBO + sum of the log prior densities of ALL trainable parameters)): on the
        # host-side hyper-parameters their gradient is added on the host every step; on the device-resident variables (Z, q_mu,
        # q_sqrt) it is evaluated and added on the device (`_device_prior`, round 5)
        self.model, self.group = model, group
        self.natgrad_gamma = None if natgrad_gamma is None else float(natgrad_gamma)
        self.mean_const = float(c)
        self.family = k.family if k is not None else None
        self.opt = _Adam(learning_rate, beta_1, beta_2, epsilon)
        # host side: unconstrained scalars (their constrained values are host arguments of the C-ABI)
        if self.sep is not None:
            self.host, self.member_names, seen = {}, [], {}
            for k_ in self.sep:                                   # (a Parameter shared by several latents' kernels: one entry, summed gradient)
                names = []
                for par, pre in ((k_.variance, "kvar"), (k_.lengthscales, "kls")):
                    if id(par) not in seen:
                        seen[id(par)] = f"{pre}_{len(seen)}"
                        self.host[seen[id(par)]] = par
                    names.append(seen[id(par)])
                self.member_names.append(tuple(names))
        elif self.combo is None:
            self.host = {"variance": k.variance, "lengthscales": k.lengthscales}
        else:
            # one host entry per distinct Parameter (a Parameter shared by several members -- k + k, tied lengthscales -- collects the
            # SUM of its members' gradients, as autodiff returns it); member i reads "kvar_<a>" / "kls_<b>"
            self.host, self.member_names, seen = {}, [], {}
            for pv, pl in self.combo[1]:
                names = []
                for par, pre in ((pv, "kvar"), (pl, "kls")):
                    if id(par) not in seen:
                        seen[id(par)] = f"{pre}_{len(seen)}"
                        self.host[seen[id(par)]] = par
                    names.append(seen[id(par)])
                self.member_names.append(tuple(names))
        # noise: a constant variance, or (a heteroskedastic Gaussian likelihood, likelihoods/scalar_continuous.py:52-111) the Parameters of
        # the noise Function -- they stay on the host like the other hyper-parameters; sigma_n^2 at the minibatch rows is formed on the
        # device every step and dF/d sigma_n^2 comes back per row, chained through the Function there (Gaussian.noise_param_grads)
        self.het = lik.is_heteroskedastic
        self.noise_pars = []
        if self.het:
            fn = lik.variance if lik.variance is not None else lik.scale       # (is_heteroskedastic: a Function)
            self.noise_pars = list(fn.parameters)
            for i, p in enumerate(self.noise_pars):
                self.host[f"noise_fn_{i}"] = p
        else:
            self.host["noise_variance"] = lik.variance
        from .mean_functions import Constant
        mf = model.mean_function
        if isinstance(mf, Constant) and hasattr(mf, "c"):   # (Zero is a Constant without a parameter, functions.py:195-204)
            # Constant.c is a trainable Parameter like any other (gpflow/functions.py:173-192): it joins the host set
            if np.size(mf.c.numpy()) != 1:
                raise NotImplementedError("the reverse pass covers a scalar Constant mean")
            self.host["mean_const"] = mf.c
        self.u = {n: np.array(p.unconstrained_variable, dtype=np.float64, copy=True) for n, p in self.host.items()}
        # device side (identity / fill-triangular transforms: the constrained array IS the variable)
        self.dev = {"Z": ops.to_device(iv.Z.numpy()).clone(), "q_mu": ops.to_device(model.q_mu.numpy()).clone(),
                    "q_sqrt": ops.to_device(model.q_sqrt.numpy()).clone()}
        self.dev_params = {"Z": iv.Z, "q_mu": model.q_mu, "q_sqrt": model.q_sqrt}
        self.q_lower = 0.0
        if self.q_diag:
            # q_sqrt [M, P] holds standard deviations under positive() = softplus (+ an optional lower bound,
            # utilities/bijectors.py:27-45): the DEVICE variable is the unconstrained one, q = softplus(u) + lower is formed
            # on the device every step and the gradient is chained through sigmoid(u) there
            from .base import Chain, Shift, Softplus
            t = model.q_sqrt.transform
            if isinstance(t, Chain) and len(t.bijectors) == 2 and isinstance(t.bijectors[0], Shift) \
                    and isinstance(t.bijectors[1], Softplus):
                self.q_lower = float(t.bijectors[0].shift)
            elif not isinstance(t, Softplus):
                raise NotImplementedError("the trainer covers the softplus transform of a diagonal q_sqrt")
            self.dev["q_sqrt"] = ops.to_device(np.asarray(model.q_sqrt.unconstrained_variable, dtype=np.float64)).clone()
        self.last_info: Optional[int] = None   # factorisation status of the last step (0 = ok), checked every step

    def constrained(self, name: str) -> np.ndarray:
        return np.asarray(self.host[name].transform.forward(self.u[name]), dtype=np.float64)

    def _device_prior(self, name: str):
        """(log prior density as a 0-d device tensor, its gradient w.r.t. the device variable) of a device-resident variable
        carrying a prior -- gpflow/base.py:201-224 evaluated where the variable lives.  Z and q_mu have identity transforms; a full
        q_sqrt is its own constrained value (fill-triangular: a permutation, unit Jacobian; only the lower triangle is a variable);
        a diagonal q_sqrt is kept unconstrained on the device, q = softplus(u) + lower."""
        from .base import PriorOn
        from .priors import grad_log_prob_device, log_prob_device
        p = self.dev_params[name]
        x = self.dev[name]
        if name == "q_sqrt" and self.q_diag:
            sig = torch.sigmoid(x)
            if p.prior_on == PriorOn.CONSTRAINED:
                q = torch.nn.functional.softplus(x) + self.q_lower
                return log_prob_device(p.prior, q), grad_log_prob_device(p.prior, q) * sig
            # prior on the unconstrained value, density reported in the constrained space: - log |dq/du| = - log sigmoid(u)
            return log_prob_device(p.prior, x) - torch.log(sig).sum(), grad_log_prob_device(p.prior, x) - (1.0 - sig)
        if name == "q_sqrt":
            low = torch.tril(x)
            if p.prior_on == PriorOn.CONSTRAINED:      # the constrained value is the whole [P, M, M] array, zeros included
                return log_prob_device(p.prior, low), torch.tril(grad_log_prob_device(p.prior, low))
            mask = torch.tril(torch.ones_like(x[0], dtype=torch.bool))
            vec = low[:, mask]                          # the unconstrained vector: the lower-triangular entries
            g = torch.zeros_like(x)
            g[:, mask] = grad_log_prob_device(p.prior, vec)
            return log_prob_device(p.prior, vec), g
        return log_prob_device(p.prior, x), grad_log_prob_device(p.prior, x)

    def step(self, data, *, global_batch: Optional[int] = None) -> torch.Tensor:
        """One Adam step on the minibatch (or this rank's shard of it); returns the objective BEFORE the update -- the ELBO
        estimate plus the log prior density of every trainable parameter that carries a prior, i.e. -training_loss
        (models/model.py:56-76), the same quantity `SVGP.elbo_and_grad` reports -- as a device tensor [1] (no host
        synchronisation beyond the scalar-gradient read-back)."""
        import torch.distributed as dist
        Xb, Yb = ops.to_device(data[0]), ops.to_device(data[1])
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        rows = int(global_batch) if global_batch is not None else Xb.shape[0] * world
        scale = 1.0 if self.model.num_data is None else float(self.model.num_data) / float(rows)
        if self.combo is None and self.sep is None:
            var = float(self.constrained("variance"))
            ls = self.constrained("lengthscales")
        if self.het:
            # SIDE EFFECT, by design: the noise Function evaluates itself from its own Parameters, so the trainer's current values of
            # them are written into the model on every step -- unlike the kernel / Z / q parameters, which reach the model only through
            # sync_to_model().  Between steps the model therefore holds the NEW noise parameters next to the kernel, Z and q of the last
            # sync: call sync_to_model() before evaluating model.elbo() / predict_* mid-training.
            for i, p in enumerate(self.noise_pars):              # the Function reads its Parameters: the trainer's current values
                p.assign_unconstrained(self.u[f"noise_fn_{i}"])
            noise = self.model.likelihood.noise_for(Xb)          # sigma_n^2 at the rows of this (shard of the) minibatch  [B]
        else:
            noise = float(self.constrained("noise_variance"))
        if "mean_const" in self.host:
            self.mean_const = float(np.ravel(self.constrained("mean_const"))[0])
        fn = gradients.svgp_elbo_and_grad if self.model.whiten else gradients.svgp_elbo_and_grad_unwhitened
        from .models.svgp import SVGP
        q_sqrt = torch.nn.functional.softplus(self.dev["q_sqrt"]) + self.q_lower if self.q_diag else self.dev["q_sqrt"]
        if self.sep is not None:
            if Yb.shape[1] != len(self.sep):
                raise ValueError(f"{len(self.sep)} separate kernels need {len(self.sep)} output columns, got {Yb.shape[1]}")
            scatter = lambda gz: gz  # noqa: E731  (each latent's input gradient is scattered below)
            F, info = None, None
            g = {"Z": torch.zeros_like(self.dev["Z"]), "q_mu": torch.zeros_like(self.dev["q_mu"]), "q_sqrt": torch.zeros_like(q_sqrt)}
            for p_, (k_, (nv_, nl_)) in enumerate(zip(self.sep, self.member_names)):
                Zs, Xs, sc = SVGP._sliced(k_, self.dev["Z"], Xb)
                Fp, gp, ip = fn(Zs, Xs, Yb[:, p_:p_ + 1].contiguous(), self.dev["q_mu"][:, p_:p_ + 1].contiguous(),
                                q_sqrt[p_:p_ + 1].contiguous(), variance=float(np.ravel(self.constrained(nv_))[0]),
                                lengthscales=self.constrained(nl_), noise_variance=noise, jitter=config.default_jitter(), scale=scale,
                                mean_const=self.mean_const, kl_weight=1.0 / world, family=k_.family)
                F = Fp if F is None else F + Fp
                info = ip if info is None else torch.maximum(info, ip)
                g["Z"] += sc(gp["Z"])
                g["q_mu"][:, p_:p_ + 1] = gp["q_mu"]
                g["q_sqrt"][p_:p_ + 1] = gp["q_sqrt"]
                for name in ("noise_variance", "mean_const"):
                    g[name] = g[name] + gp[name] if name in g else gp[name]
                gvp, glp = gp["variance"].reshape(1), gp["lengthscales"].reshape(-1)
                g[nv_] = g[nv_] + gvp if nv_ in g else gvp
                g[nl_] = g[nl_] + glp if nl_ in g else glp
        elif self.combo is None:
            Zs, Xs, scatter = SVGP._sliced(self.kernel, self.dev["Z"], Xb)      # active_dims (kernels/base.py:90-109)
            F, g, info = fn(
                Zs, Xs, Yb, self.dev["q_mu"], q_sqrt, variance=var, lengthscales=ls,
                noise_variance=noise, jitter=config.default_jitter(), scale=scale, mean_const=self.mean_const,
                kl_weight=1.0 / world, family=self.family)
            g = dict(g)
        else:
            spec0 = self.combo[0]
            members = [(f, float(np.ravel(self.constrained(nv))[0]), self.constrained(nl))
                       for (f, _, _), (nv, nl) in zip(spec0.members, self.member_names)]
            spec = gradients.KernelSpec(members, spec0.tree if spec0.tree is not None else spec0.op, spec0.cols)
            scatter = lambda gz: gz  # noqa: E731  (the spec slices for its members and scatters their input gradients itself)
            F, g, info = fn(self.dev["Z"], Xb.contiguous(), Yb, self.dev["q_mu"], q_sqrt, noise_variance=noise,
                            jitter=config.default_jitter(), scale=scale, mean_const=self.mean_const, kl_weight=1.0 / world,
                            kernel_spec=spec)
            g = dict(g)
            gv, gl = g.pop("variance"), g.pop("lengthscales")
            if spec.n == 1:
                gv, gl = gv.reshape(1), [gl]
            for i, (nv, nl) in enumerate(self.member_names):    # per-member gradients onto their (possibly shared) Parameters
                g[nv] = g[nv] + gv[i].reshape(1) if nv in g else gv[i].reshape(1)
                gli = gl[i].reshape(-1)
                g[nl] = g[nl] + gli if nl in g else gli
        if self.het:
            # per-row dF/d sigma_n^2 -> the noise Function's parameters (this shard's rows; summed over the ranks below)
            rows_g = g.pop("noise_variance")
            acc = {}
            for par, gv in self.model.likelihood.noise_param_grads(Xb, rows_g):
                acc[id(par)] = acc[id(par)] + gv if id(par) in acc else gv
            for i, p in enumerate(self.noise_pars):
                g[f"noise_fn_{i}"] = acc[id(p)].reshape(-1).contiguous() if id(p) in acc else \
                    torch.zeros(int(np.size(self.u[f"noise_fn_{i}"])), dtype=torch.float64, device=Xb.device)
        g["Z"] = scatter(g["Z"])
        if self.q_diag:
            g["q_sqrt"] = g["q_sqrt"] * torch.sigmoid(self.dev["q_sqrt"])      # d softplus(u) / du
        g["_status"] = info.to(torch.float64).reshape(-1)[:1]   # rides in the packed all-reduce: > 0 iff ANY rank failed
        F, g = distributed.all_reduce_grads(F, g, self.group)
        # The factorisation status decides whether this step may be applied at all: after a failed Cholesky of Kuu the
        # gradients are garbage, and applying them would poison the variables AND the Adam moments for good (the
        # reference raises from tf.linalg.cholesky at this point).  It rides in the step's one read-back and, with
        # several ranks, in the packed gradient all-reduce (summed), so every rank takes the same decision.
        status = g.pop("_status")
        hnames = list(self.host)                                 # variance, lengthscales, noise (constant or Function parameters), [mean]
        hsizes = [1 if n == "mean_const" else int(g[n].numel()) for n in hnames]
        small = torch.cat([g[n].reshape(-1)[:sz] for n, sz in zip(hnames, hsizes)] + [status])
        small = small.cpu().numpy()                              # the step's one read-back: 4 + |lengthscales| doubles (+ the noise Function's)
        self.last_info = int(small[-1])
        if self.last_info != 0:
            from ._lib import GpkError
            # (summed over the ranks: pivot columns are far below 2^31, so a sum that reaches INT_MAX contains a timed-out hand-off)
            if self.last_info >= ops.INFO_HANDOFF_TIMEOUT:
                raise GpkError("the factorisation of Kuu failed: an internal stream hand-off timed out (status INT_MAX)" +
                               ("" if world == 1 else " on at least one rank") + "; the step was NOT applied")
            raise GpkError("Cholesky decomposition of Kuu was not successful" +
                           (f" (non-positive pivot at column {self.last_info - 1})" if world == 1 else
                            " on at least one rank") + "; the step was NOT applied")
        adam_names = ("Z", "q_mu", "q_sqrt")
        nat = None
        if self.natgrad_gamma is not None:
            # into temporaries first: natgrad_update has a status check of its own ("precision (step too long?)"); if it
            # raises, neither q(u), nor the Adam step counter, nor any moment has been touched
            from . import natgrad
            adam_names = ("Z",)
            nat = natgrad.natgrad_update(self.dev["q_mu"], self.dev["q_sqrt"], -g["q_mu"], -g["q_sqrt"], self.natgrad_gamma)
        # MAP (model.py:47-76): the priors of the device-resident variables, evaluated BEFORE anything is updated
        log_prior = 0.0
        for name, par in self.dev_params.items():
            if par.prior is not None and par.trainable:
                if name != "Z" and nat is not None:
                    raise NotImplementedError("a prior on q_mu / q_sqrt together with natural-gradient steps on q(u)")
                lp, gp = self._device_prior(name)
                g[name] = g[name] + gp
                log_prior = log_prior + lp
        self.opt.t += 1
        if nat is not None:
            self.dev["q_mu"].copy_(nat[0])
            self.dev["q_sqrt"].copy_(nat[1])
        for name in adam_names:                                  # minimise -F
            if self.dev_params[name].trainable:
                self.opt.update_device(name, self.dev[name], -g[name])
        offs = np.concatenate([[0], np.cumsum(hsizes)])
        parts = {n: small[offs[i]:offs[i + 1]] for i, n in enumerate(hnames)}
        for name, p in self.host.items():
            if not p.trainable:
                continue
            gu = -parts[name].reshape(self.u[name].shape) * p.transform.forward_grad(self.u[name])
            if p.prior is not None:   # loss = -(ELBO + log prior): the prior's part, evaluated at the trainer's current value
                p.assign_unconstrained(self.u[name])
                gu = gu - p.log_prior_density_grad()
                log_prior = log_prior + p.log_prior_density()
            self.u[name] = self.opt.update_host(name, self.u[name], gu)
        return F + log_prior

    def sync_to_model(self) -> None:
        for name, p in self.host.items():
            p.assign_unconstrained(self.u[name])
        for name, p in self.dev_params.items():
            v = self.dev[name].cpu().numpy()
            if name == "q_sqrt" and self.q_diag:
                p.assign_unconstrained(v)
            else:
                p.assign(np.tril(v) if name == "q_sqrt" else v)
