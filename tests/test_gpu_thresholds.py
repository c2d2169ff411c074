"""GPU: parity on BOTH sides of every schedule switch of the factorisation driver (potrf_core: ~10 schedules chosen on row-count / M /
workgroup-count thresholds that were tuned on the bench shapes).  The reference has no shape preference
(conditionals/util.py:84-169, gpr.py:91-107): whatever schedule runs, the ELBO / the factor must be the reference's.
The checker is the torch-fp64 restatement of the reference (oracle/gp_oracle_grad.py: cholesky + solve_triangular + dense L_q^T A),
multi-threaded, so that M = 2048 shapes stay cheap."""
import numpy as np
import pytest
import torch

from oracle import gp_oracle_grad as orct

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import gpflow_amd
    return gpflow_amd


# (M, rows): 6144 rows switch three extra-row-stream schedules (progressive first group, tiled rest-updates -- which additionally
# switch at 150 workgroups, crossed panel by panel inside every M = 2048 run --, no shrinking groups); M <= 1024 changes the width
# of the extra-row groups; 256 extra rows separate "ride along" from the right-looking row solve; 1024 rows the fused in-group solve
@pytest.mark.parametrize("m,rows", [(2048, 6143), (2048, 6144), (2048, 6145), (1024, 6144), (1152, 6144), (1024, 1000), (1152, 1000),
                                    (1024, 1023), (1024, 1025), (640, 256), (640, 257), (2048, 300)])
def test_elbo_on_both_sides_of_every_schedule_switch(gp, m, rows):
    from gpflow_amd import ops
    rng = np.random.default_rng(m + rows)
    d = 4
    X = rng.normal(size=(rows, d)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(rows, 1))
    Z = rng.normal(size=(m, d)); q_mu = 0.1 * rng.normal(size=(m, 1))
    q_sqrt = np.tril(0.05 * rng.normal(size=(1, m, m))) + 0.5 * np.eye(m)
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    out, info = ops.svgp_elbo_shard(ops.to_device(Z), ops.to_device(X), ops.to_device(Y), ops.to_device(q_mu), ops.to_device(q_sqrt),
                                    variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6)
    ops.check_info(info)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    with torch.no_grad():
        ref = float(orct.svgp_elbo_torch(t(X), t(Y), t(Z), t(q_mu), t(q_sqrt), torch.tensor(1.0, dtype=torch.float64), t(ls),
                                         torch.tensor(0.1, dtype=torch.float64), num_data=None, whiten=True))
    o = out.cpu().numpy()
    got = float(o[0] - o[1])   # sum of the variational expectations - KL  (num_data None: scale 1)
    assert abs(got - ref) <= 1e-9 * abs(ref), (m, rows, got, ref)


# 4096 switches from single-leaf panels to 640-column outer panels with a CU-masked bulk stream and a 4096-column narrow tail
@pytest.mark.parametrize("n", [4095, 4096, 4097])
def test_factor_on_both_sides_of_the_outer_panel_switch(gp, n):
    from gpflow_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n)
    B = torch.randn(n, 64, generator=g, dtype=torch.float64)
    K = B @ B.T / 64 + torch.eye(n, dtype=torch.float64) * 2.0
    T = ops.to_device(K.numpy())
    _, info = ops.potrf_(T, n, zero_upper=True)
    ops.check_info(info)
    L = T.cpu()
    resid = (L @ L.T - K).abs().max().item()
    assert resid <= 1e-12 * n, (n, resid)
