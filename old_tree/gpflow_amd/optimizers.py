"""`gpflow.optimizers.Scipy` (gpflow/optimizers/scipy.py:45-331) for the models whose objective has a hand-written
gradient on the device: the reference packs the trainable variables' UNCONSTRAINED values into one vector
(`scipy.py:289-305`), evaluates loss and gradient with TF (`:322-331`) and hands both to `scipy.optimize.minimize`;
here loss and gradient come from `model.objective_and_grad()` (GPR, SGPR; gpflow_amd/gradients.py)."""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import scipy.optimize


class Scipy:
    def minimize(self, model, data=None, *, method: str = "L-BFGS-B", options: Optional[Dict[str, Any]] = None,
                 **scipy_kwargs) -> scipy.optimize.OptimizeResult:
        """Minimise the training loss (-LML of a GPR, -ELBO of an SGPR, -ELBO of an SVGP on the fixed batch `data`) over
        the model's trainable parameters; the model holds the optimum afterwards."""
        if data is not None and hasattr(model, "elbo_and_grad"):
            objective = lambda: model.elbo_and_grad(data)  # noqa: E731
        elif hasattr(model, "objective_and_grad"):
            objective = model.objective_and_grad
        else:
            raise NotImplementedError(f"{type(model).__name__} has no device gradient (SVGP needs `data`)")
        _, g0 = objective()
        params = list(g0)
        sizes = [int(np.size(p.unconstrained_variable)) for p in params]

        def unpack(x):
            off = 0
            for p, n in zip(params, sizes):
                p.assign_unconstrained(np.asarray(x[off:off + n]).reshape(np.shape(p.unconstrained_variable)))
                off += n

        def fun(x):
            unpack(x)
            try:
                v, g = objective()
            except Exception as e:  # a failed factorisation during a line search: reject the point, as scipy expects
                if "not successful" not in str(e):
                    raise
                return 1e300, np.zeros_like(x)
            return -v, -np.concatenate([np.ravel(g[p]) for p in params])

        x0 = np.concatenate([np.ravel(p.unconstrained_variable) for p in params]).astype(np.float64)
        res = scipy.optimize.minimize(fun, x0, jac=True, method=method, options=options or {}, **scipy_kwargs)
        unpack(res.x)
        return res


class NaturalGradient:
    """`gpflow.optimizers.NaturalGradient(gamma)` (natgrad.py:155-368) for the (q_mu, q_sqrt) of an SVGP, natural
    parametrisation.  The reference takes a loss closure and differentiates it with TF; here the loss is the model's
    -ELBO on `data` and its gradient comes from the device reverse pass (gradients.svgp_elbo_and_grad), so the call is

        NaturalGradient(gamma=1.0).minimize(model, data)      # one step; updates model.q_mu / model.q_sqrt

    (SVGP, whitened or not, SquaredExponential / Matern kernel, Gaussian likelihood, full q_sqrt -- the scope of the reverse pass)."""

    def __init__(self, gamma: float = 1.0, xi_transform: str = "XiNat"):
        """xi_transform: "XiNat" (natural parameters, the default of the reference) or "XiSqrtMeanVar" (steps taken in
        (q_mu, q_sqrt) itself, natgrad.py:139-173); objects named like the reference's classes are accepted too."""
        from . import natgrad
        self.gamma = float(gamma)
        name = xi_transform if isinstance(xi_transform, str) else type(xi_transform).__name__
        if name not in natgrad.XI_TRANSFORMS:
            raise NotImplementedError(f"xi_transform {name!r}: only {natgrad.XI_TRANSFORMS}")
        self.xi_transform = name

    def minimize(self, model, data) -> None:
        from . import config, gradients, natgrad, ops
        k, iv, c = model.gradient_config()
        lik = model.likelihood
        X, Y = ops.to_device(data[0]), ops.to_device(data[1])
        scale = 1.0 if model.num_data is None else float(model.num_data) / float(X.shape[0])
        family, var, ls = k.hyper()
        q_mu, q_sqrt = model.q_mu.device_value(), model.q_sqrt.device_value()
        fn = gradients.svgp_elbo_and_grad if model.whiten else gradients.svgp_elbo_and_grad_unwhitened
        _, g, info = fn(iv.Z.device_value(), X, Y, q_mu, q_sqrt, variance=var, lengthscales=ls,
                                                  noise_variance=lik.noise_variance(), jitter=config.default_jitter(),
                                                  scale=scale, mean_const=float(c), family=family)
        ops.check_info(info)
        mu, sq = natgrad.natgrad_update(q_mu, q_sqrt, -g["q_mu"], -g["q_sqrt"], self.gamma,
                                        xi_transform=self.xi_transform)   # loss = -ELBO
        model.q_mu.assign(mu.cpu().numpy())
        model.q_sqrt.assign(sq.cpu().numpy())
