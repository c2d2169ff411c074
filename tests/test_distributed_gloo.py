"""CPU, world_size 2, gloo: the data-parallel ELBO protocol (shard rows, one all-reduce of the per-shard
data term, replicated KL) reproduces the single-process ELBO.  The per-shard device computation is
replaced by the oracle here (no GPU in this container) -- the thing under test is the sharding /
collective / scaling logic of gpflow_amd.distributed, which is backend-agnostic (RCCL on the GPUs)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gpflow_amd.distributed import sharded_elbo, shard_bounds
        from oracle import gp_oracle as orc
        rng = np.random.default_rng(11)  # identical data on every rank (replicated minibatch)
        B, M, D, P = 101, 9, 2, 2
        X = rng.normal(size=(B, D)); Y = rng.normal(size=(B, P)); Z = rng.normal(size=(M, D))
        q_mu = rng.normal(size=(M, P)); q_sqrt = np.tril(rng.normal(size=(P, M, M))) * 0.2 + np.eye(M)
        kw = dict(variance=1.1, lengthscales=0.9, noise_variance=0.2)
        calls = []

        def local_terms(lo, hi):
            calls.append((lo, hi))
            s, kl = orc.svgp_elbo_terms(X[lo:hi], Y[lo:hi], Z, q_mu, q_sqrt, **kw)
            return torch.tensor([s, kl], dtype=torch.float64)

        elbo = float(sharded_elbo(local_terms, B, num_data=5000))
        ref = orc.svgp_elbo(X, Y, Z, q_mu, q_sqrt, num_data=5000, **kw)
        q.put((rank, elbo, ref, calls[0], shard_bounds(B, world, rank)))
    finally:
        dist.destroy_process_group()


def test_sharded_elbo_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    (r0, e0, ref0, c0, s0), (r1, e1, ref1, c1, s1) = results
    assert c0 == s0 == (0, 51) and c1 == s1 == (51, 101)  # disjoint cover of the minibatch
    assert e0 == e1  # every rank ends with the same ELBO
    np.testing.assert_allclose(e0, ref0, rtol=1e-12)


def test_sharded_elbo_single_process():
    from gpflow_amd.distributed import sharded_elbo
    out = sharded_elbo(lambda lo, hi: torch.tensor([float(hi - lo), 2.0], dtype=torch.float64), 10, num_data=100)
    assert float(out) == pytest.approx(10 * 10.0 - 2.0)


def _train_worker(rank, world, port, q):
    """Data-parallel TRAINING step: each rank evaluates value + gradient on its row shard (emulated primitives: no GPU
    here), one all-reduce of the packed gradient, identical Adam update everywhere."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fake_ops
        import gpflow_amd as gpflow
        from gpflow_amd import gradients, training
        from gpflow_amd.distributed import shard_bounds
        from oracle import gp_oracle_grad as orcg
        gradients.ops = fake_ops
        training.ops = fake_ops
        rng = np.random.default_rng(21)  # identical on every rank
        B, M, D, P = 90, 24, 2, 1
        X = rng.normal(size=(B, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(B, P))
        Z = rng.normal(size=(M, D)); q_mu = 0.2 * rng.normal(size=(M, P))
        q_sqrt = np.tril(0.05 * rng.normal(size=(P, M, M))) + 0.7 * np.eye(M)
        model = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.2, lengthscales=[0.9, 1.1]),
                                   gpflow.likelihoods.Gaussian(0.3), Z.copy(), q_mu=q_mu.copy(), q_sqrt=q_sqrt.copy(),
                                   num_data=4000)
        v, g = orcg.svgp_elbo_value_and_grads(X, Y, Z, q_mu, q_sqrt, variance=1.2, lengthscales=[0.9, 1.1],
                                              noise_variance=0.3, num_data=4000)
        tr = training.SVGPTrainer(model, learning_rate=1e-2)
        lo, hi = shard_bounds(B, world, rank)
        F = tr.step((X[lo:hi], Y[lo:hi]), global_batch=B)
        tr.sync_to_model()
        lr_t = 1e-2 * np.sqrt(1 - 0.999) / (1 - 0.9)
        gl = -g["q_mu"]
        expect_qmu = q_mu - lr_t * (0.1 * gl) / (np.sqrt(0.001 * gl ** 2) + 1e-7)
        q.put((rank, float(F[0]), v, float(np.abs(model.q_mu.numpy() - expect_qmu).max()),
               model.kernel.lengthscales.numpy().tolist()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_training_step_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    (_, f0, v0, err0, ls0), (_, f1, v1, err1, ls1) = results
    assert f0 == f1 and ls0 == ls1                 # every rank holds the same objective and parameters afterwards
    assert abs(f0 - v0) <= 1e-10 * abs(v0)          # sum of the shard objectives == full-batch ELBO (oracle)
    assert err0 <= 1e-9 and err1 <= 1e-9            # Adam step on the all-reduced gradient == step on the full gradient


def _sgpr_worker(rank, world, port, q):
    """SGPR over row shards: per-shard statistics (emulated primitives), ONE all-reduce of the packed M x M + M x P + 2
    buffer, replicated tail."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fake_ops
        from gpflow_amd import gradients
        from gpflow_amd.distributed import shard_bounds
        from gpflow_amd.models import sgpr
        from oracle import gp_oracle as orc
        gradients.ops = fake_ops
        sgpr.ops = fake_ops
        rng = np.random.default_rng(31)
        N, M, D, P = 301, 40, 2, 2
        X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P)); Z = rng.normal(size=(M, D))
        t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))  # noqa: E731
        lo, hi = shard_bounds(N, world, rank)
        _, _, packed = sgpr.shard_statistics(t(Z), t(X[lo:hi]), t(Y[lo:hi]), variance=1.1, lengthscales=0.8,
                                             family="SquaredExponential", jitter=1e-6, mean_const=0.0)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        elbo = float(sgpr.elbo_from_statistics(packed, M, P, N, variance=1.1, noise_variance=0.3))
        ref = orc.sgpr_elbo(X, Y, Z, variance=1.1, lengthscales=0.8, noise_variance=0.3)
        q.put((rank, elbo, ref))
    finally:
        dist.destroy_process_group()


def test_sgpr_row_sharded_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sgpr_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, e0, ref0), (_, e1, _) = results
    assert e0 == e1
    assert abs(e0 - ref0) <= 1e-10 * abs(ref0)


def _sgpr_grad_worker(rank, world, port, q):
    """Gradients of the row-sharded SGPR through the MODEL surface (SGPR(sharded=True).objective_and_grad): every rank holds
    the complete ELBO and gradient after two all-reduces; rank 1 of the second case holds an EMPTY shard."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fake_ops
        import gpflow_amd as gpflow
        from gpflow_amd import ops
        from gpflow_amd.distributed import shard_bounds
        from oracle import gp_oracle_grad as orcg
        for name in dir(fake_ops):
            if not name.startswith("_") and callable(getattr(fake_ops, name)) and hasattr(ops, name) \
                    and name not in ("torch", "np", "sla"):
                setattr(ops, name, getattr(fake_ops, name))
        rng = np.random.default_rng(41)
        N, M, D, P = 257, 36, 3, 2
        X = rng.normal(size=(N, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(N, P)); Z = rng.normal(size=(M, D))
        ls = np.array([0.8, 1.1, 1.4])
        out = []
        for family, bounds in (("SquaredExponential", shard_bounds(N, world, rank)), ("Matern52", (0, N) if rank == 0 else (N, N))):
            lo, hi = bounds
            kern = getattr(gpflow.kernels, family)(variance=1.2, lengthscales=ls)
            m = gpflow.models.SGPR((X[lo:hi], Y[lo:hi]), kern, Z.copy(), noise_variance=0.3, sharded=True,
                                   mean_function=gpflow.mean_functions.Constant(0.2))
            v, g = m.objective_and_grad()
            rv, rg = orcg.sgpr_elbo_value_and_grads(X, Y, Z, variance=1.2, lengthscales=ls, noise_variance=0.3, mean=0.2, family=family)
            errs = {"value": abs(v - rv) / abs(rv), "elbo": abs(v - float(m.elbo())) / abs(rv),
                    "Z": float(np.abs(np.asarray(g[m.inducing_variable.Z]) - rg["Z"]).max() / np.abs(rg["Z"]).max()),
                    "mean": float(abs(np.ravel(g[m.mean_function.c])[0] - rg["mean_const"]) / max(1.0, abs(float(rg["mean_const"]))))}
            for par, name in ((kern.variance, "variance"), (kern.lengthscales, "lengthscales"), (m.likelihood.variance, "noise_variance")):
                u = par.unconstrained_variable
                ref = np.ravel(rg[name]) * np.ravel(par.transform.forward_grad(u))
                errs[name] = float(np.abs(np.ravel(g[par]) - ref).max() / max(1.0, np.abs(ref).max()))
            out.append((family, v, errs))
        # a heteroskedastic likelihood on the row-sharded model: dF/d sigma_n^2 stays with the rows of the shard, the noise function's
        # parameter gradients are summed over the ranks; the upper bound's second statistics pass needs the GLOBAL c of the first
        from oracle import gp_oracle as orc
        Xp = rng.random((N, 2)); Yp = np.sin(5 * Xp[:, :1]) + (0.7 - 0.5 * Xp[:, :1]) * rng.standard_normal((N, 1)); Zp = Xp[:M].copy()
        A0, b0 = np.array([[-0.3], [0.05]]), np.array([0.6])
        rv, rg = orcg.heteroskedastic_value_and_grads("sgpr", Xp, Yp, A=A0, b=b0, variance=1.1, lengthscales=[0.25, 0.9], Z=Zp)
        nv = np.maximum(Xp @ A0 + b0, 1e-3)[:, 0] ** 2
        rub = orc.sgpr_upper_bound(Xp, Yp, Zp, variance=1.1, lengthscales=np.array([0.25, 0.9]), noise_variance=nv)
        # ... also with a rank that holds ONE row and one that holds NONE: a per-row noise vector with one entry (or none) is still a
        # per-row vector -- the reverse pass once took the constant-noise branch for it: inconsistent ELBOs across the ranks, and a
        # rank without rows raised before the all-reduce the others were waiting in
        for tag, (lo, hi) in (("heteroskedastic", shard_bounds(N, world, rank)),
                              ("heteroskedastic, one row on rank 1", (0, N - 1) if rank == 0 else (N - 1, N)),
                              ("heteroskedastic, no rows on rank 1", (0, N) if rank == 0 else (N, N))):
            mh = gpflow.models.SGPR((Xp[lo:hi], Yp[lo:hi]), gpflow.kernels.SquaredExponential(variance=1.1, lengthscales=[0.25, 0.9]), Zp.copy(),
                                    likelihood=gpflow.likelihoods.Gaussian(scale=gpflow.functions.Linear(A=A0.copy(), b=b0.copy())), sharded=True)
            v, g = mh.objective_and_grad()
            ub = float(mh.upper_bound())
            errs = {"value": abs(v - rv) / abs(rv), "upper": abs(ub - rub) / abs(rub),
                    "A": float(np.abs(np.asarray(g[mh.likelihood.scale.A]).reshape(rg["A"].shape) - rg["A"]).max() / max(1.0, np.abs(rg["A"]).max())),
                    "b": float(np.abs(np.ravel(g[mh.likelihood.scale.b]) - np.ravel(rg["b"])).max() / max(1.0, np.abs(rg["b"]).max())),
                    "Z": float(np.abs(np.asarray(g[mh.inducing_variable.Z]) - rg["Z"]).max() / np.abs(rg["Z"]).max())}
            out.append((tag, v, errs))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_sgpr_row_sharded_gradients_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sgpr_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, out0), (_, out1) = results
    for (fam, v0, e0), (_, v1, e1) in zip(out0, out1):
        assert v0 == v1, fam                                   # both ranks hold the same ELBO
        for errs in (e0, e1):
            for name, e in errs.items():
                assert e <= 1e-8, (fam, name, e)


def _product_worker(rank, world, port, q, backend):
    """`distributed.svgp_elbo_data_parallel` on the PRODUCT host code (SVGP model surface -> elbo_terms -> fused shard):
    gloo + emulated primitives on CPU, nccl (= RCCL) + the HIP library when GPUs are visible."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gpflow_amd as gpflow
        from gpflow_amd import distributed, ops
        from oracle import gp_oracle as orc
        if backend == "gloo":
            import fake_ops
            for name in dir(fake_ops):
                if not name.startswith("_") and callable(getattr(fake_ops, name)) and hasattr(ops, name) \
                        and name not in ("torch", "np", "sla"):
                    setattr(ops, name, getattr(fake_ops, name))
        rng = np.random.default_rng(21)  # identical on every rank
        B, M, D, P = 301, 140, 3, 2
        X = rng.normal(size=(B, D)); Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(B, P))
        Z = rng.normal(size=(M, D)); q_mu = 0.2 * rng.normal(size=(M, P))
        q_sqrt = np.tril(rng.normal(size=(P, M, M))) * 0.05 + 0.5 * np.eye(M)
        ls = 0.9 * np.sqrt(D)
        m = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.1, lengthscales=ls),
                               gpflow.likelihoods.Gaussian(0.2), Z, q_mu=q_mu, q_sqrt=q_sqrt, num_data=7000)
        elbo = float(distributed.svgp_elbo_data_parallel(m, (ops.to_device(X), ops.to_device(Y))))
        ref = orc.svgp_elbo(X, Y, Z, q_mu, q_sqrt, variance=1.1, lengthscales=ls, noise_variance=0.2, num_data=7000)
        # the same exchange with one kernel per latent (gpk_svgp_elbo_shard_sep behind SVGP.elbo_terms)
        vs, lss = [1.1, 0.8], [0.9 * np.sqrt(D), 1.2 * np.sqrt(D)]
        ksep = gpflow.kernels.SeparateIndependent([gpflow.kernels.SquaredExponential(variance=v, lengthscales=l)
                                                   for v, l in zip(vs, lss)])
        iv = gpflow.inducing_variables.SharedIndependentInducingVariables(gpflow.inducing_variables.InducingPoints(Z))
        ms = gpflow.models.SVGP(ksep, gpflow.likelihoods.Gaussian(0.2), iv, q_mu=q_mu, q_sqrt=q_sqrt, num_latent_gps=P,
                                num_data=7000)
        assert ms._fused_separate_config() is not None
        elbo_s = float(distributed.svgp_elbo_data_parallel(ms, (ops.to_device(X), ops.to_device(Y))))
        ref_s = orc.svgp_elbo_separate(X, Y, [Z] * P, q_mu, q_sqrt, variances=vs, lengthscales_list=lss, noise_variance=0.2,
                                       whiten=True, num_data=7000)
        np.testing.assert_allclose(elbo_s, ref_s, rtol=1e-9)
        # and un-whitened (gpk_svgp_elbo_shard(whiten = 0): the KL is replicated, the data term all-reduced, as above)
        mu = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.1, lengthscales=ls), gpflow.likelihoods.Gaussian(0.2), Z,
                                q_mu=q_mu, q_sqrt=q_sqrt, num_data=7000, whiten=False)
        assert mu._fused_config() is not None
        elbo_u = float(distributed.svgp_elbo_data_parallel(mu, (ops.to_device(X), ops.to_device(Y))))
        ref_u = orc.svgp_elbo(X, Y, Z, q_mu, q_sqrt, variance=1.1, lengthscales=ls, noise_variance=0.2, num_data=7000, whiten=False)
        np.testing.assert_allclose(elbo_u, ref_u, rtol=1e-9)
        q.put((rank, elbo, ref))
    finally:
        dist.destroy_process_group()


def _run_product(backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_product_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, e0, ref0), (_, e1, _) = results
    assert e0 == e1                                   # the all-reduce leaves every rank with the same ELBO
    np.testing.assert_allclose(e0, ref0, rtol=1e-9)   # and it is the single-process ELBO of the whole minibatch


def test_product_data_parallel_elbo_world2_gloo():
    _run_product("gloo")


@pytest.mark.gpu
def test_product_data_parallel_elbo_world2_nccl():
    """The same through RCCL on two MI355X (one process per GPU).  Skipped on boxes with fewer than 2 devices -- the
    1-GPU gpurun boxes of this build never ran it; it is here for the 8-GPU node."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 HIP devices")
    _run_product("nccl")
