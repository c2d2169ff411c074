"""`gpflow.optimizers.Scipy` (gpflow/optimizers/scipy.py:45-331) for the models whose objective has a hand-written
gradient on the device: the reference packs the trainable variables' UNCONSTRAINED values into one vector
(`scipy.py:289-305`), evaluates loss and gradient with TF (`:322-331`) and hands both to `scipy.optimize.minimize`;
here loss and gradient come from `model.log_marginal_likelihood_and_grad()` (gpflow_amd/gradients.py)."""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import scipy.optimize


class Scipy:
    def minimize(self, model, *, method: str = "L-BFGS-B", options: Optional[Dict[str, Any]] = None,
                 **scipy_kwargs) -> scipy.optimize.OptimizeResult:
        """Minimise -LML of `model` (a GPR) over its trainable parameters; the model holds the optimum afterwards."""
        if not hasattr(model, "log_marginal_likelihood_and_grad"):
            raise NotImplementedError(f"{type(model).__name__} has no device gradient; use training.SVGPTrainer for SVGP")
        _, g0 = model.log_marginal_likelihood_and_grad()
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
                v, g = model.log_marginal_likelihood_and_grad()
            except Exception as e:  # a failed factorisation during a line search: reject the point, as scipy expects
                if "not successful" not in str(e):
                    raise
                return 1e300, np.zeros_like(x)
            return -v, -np.concatenate([np.ravel(g[p]) for p in params])

        x0 = np.concatenate([np.ravel(p.unconstrained_variable) for p in params]).astype(np.float64)
        res = scipy.optimize.minimize(fun, x0, jac=True, method=method, options=options or {}, **scipy_kwargs)
        unpack(res.x)
        return res
