"""CPU: host-side logic of the GPflow surface (parameters, transforms, config, argument checking, error
behaviour) and the C-ABI library itself (loads, exports every symbol include/gpk.h declares).  No
compute calls: there is no GPU here and gpflow_amd has no CPU path."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def built_lib():
    from gpflow_amd import _lib
    if not os.path.exists(_lib.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, "include", "gpk.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gpk_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    from gpflow_amd import _lib
    for name in declared:
        assert hasattr(built_lib, name), f"libgpk.so lacks {name} declared in include/gpk.h"
    # and the ctypes table binds exactly the declared interface
    assert set(_lib.EXPORTED_SYMBOLS) == declared
    assert built_lib.gpk_version().decode().startswith("gpk")
    assert built_lib.gpk_invd_elems(300, 2) == 2 * 3 * 128 * 128
    assert built_lib.gpk_svgp_elbo_workspace_bytes(2048, 8192, 8, 1, 0, 1) > (2048 + 8192) * 2048 * 8
    # the separate-kernel driver keeps one trapezoid PER latent (config C5: 4 x (1024 + 8192) x 1024 doubles) + its tails
    sep = built_lib.gpk_svgp_elbo_sep_workspace_bytes(1024, 8192, 8, 4)
    assert 4 * (1024 + 8192) * 1024 * 8 < sep < 2 * 4 * (1024 + 8192) * 1024 * 8
    # argument errors are reported before anything touches a device (GPK_E_ARG = -1)
    assert built_lib.gpk_svgp_elbo_shard_sep(None, None, None, 1024, 8, 0, None, None, 8192, 8, 4, 8, 4, None, 1, None, 0.1, None, 1e-6,
                                             0.0, None, None, None, None, None, 0) == -1
    # an un-whitened diagonal q_sqrt keeps P + m more trapezoid rows and the second-solve buffer; the whitened one neither
    assert built_lib.gpk_svgp_elbo_workspace_bytes(2048, 8192, 8, 1, 1, 0) > built_lib.gpk_svgp_elbo_workspace_bytes(2048, 8192, 8, 1, 1, 1) \
        + 8192 * 2048 * 8
    # the single-launch step kernel of round 4 reserves nothing in the product library (its regions doubled this figure)
    assert built_lib.gpk_svgp_elbo_workspace_bytes(2048, 8192, 8, 1, 0, 1) < 1.45 * (2048 + 8192 + 2048) * 2048 * 8


def test_header_is_plain_c_and_a_c_client_links(built_lib, tmp_path):
    """The drop-in boundary is a C ABI: include/gpk.h must compile as C99 (and C++), and a C client that references
    every declared entry point must link against libgpk.so (no C++ types or mangled names leak through)."""
    import shutil
    import subprocess
    from gpflow_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "gpk.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++", hdr], check=True)
    names = sorted(_lib.EXPORTED_SYMBOLS)
    src = tmp_path / "client.c"
    src.write_text('#include "gpk.h"\n#include <stdio.h>\nint main(void) {\n  const void* f[] = {' +
                   ", ".join(f"(const void*){n}" for n in names) +
                   '};\n  printf("%s %d\\n", gpk_version(), (int)(sizeof f / sizeof f[0]));\n  return f[0] == 0;\n}\n')
    exe = tmp_path / "client"
    libdir = os.path.dirname(_lib.lib_path())
    subprocess.run(["gcc", "-std=gnu99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir,
                    "-l:" + os.path.basename(_lib.lib_path()), "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.startswith("gpk") and out.split()[-1] == str(len(names))


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from gpflow_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_no_gpu_means_error_not_fallback(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gpflow_amd as gpflow
    from gpflow_amd._lib import GpkError
    m_kernel = gpflow.kernels.RBF()
    with pytest.raises(GpkError, match="no CPU path"):
        m_kernel(np.zeros((3, 1)))
    from gpflow_amd import ops
    with pytest.raises(GpkError):
        ops.kernel_matrix(torch.zeros(3, 1, dtype=torch.float64), None, variance=1.0, lengthscales=1.0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under gpflow_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gpflow_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "gp_oracle" not in src, f


def test_parameter_transforms():
    from gpflow_amd.base import Parameter, positive, triangular
    from oracle import gp_oracle as orc
    p = Parameter(2.5, transform=positive())
    np.testing.assert_allclose(p.numpy(), 2.5)
    np.testing.assert_allclose(p.unconstrained_variable, orc.softplus_inverse(2.5))
    p.assign(0.3)
    np.testing.assert_allclose(p.numpy(), 0.3, rtol=1e-15)
    q = Parameter(1.0, transform=positive(lower=1e-6))
    np.testing.assert_allclose(q.unconstrained_variable, orc.positive_inverse(1.0, 1e-6))
    # tests/gpflow/test_base.py:30-37: a value at/below the lower bound is rejected
    with pytest.raises(ValueError, match="incompatible with this parameter's transform"):
        Parameter(0.0, transform=positive(lower=1e-6))
    with pytest.raises(ValueError):
        q.assign(-1.0)
    L = np.tril(np.arange(1.0, 10.0).reshape(3, 3))[None]
    t = Parameter(L + np.triu(np.ones((1, 3, 3)), 1), transform=triangular())  # upper part dropped
    np.testing.assert_array_equal(t.numpy(), L)
    np.testing.assert_array_equal(t.unconstrained_variable, orc.fill_triangular_inverse(L))
    e = Parameter(2.0, transform=positive(base="exp"))
    np.testing.assert_allclose(e.unconstrained_variable, np.log(2.0))
    pp = Parameter(p)
    assert pp.transform is p.transform and pp.trainable


def test_config():
    from gpflow_amd import config
    assert config.default_float() is np.float64
    assert config.default_jitter() == 1e-6
    assert config.default_likelihood_positive_minimum() == 1e-6
    assert config.default_positive_bijector() == "softplus"
    with config.as_context():
        config.set_default_jitter(1e-3)
        assert config.default_jitter() == 1e-3
    assert config.default_jitter() == 1e-6
    with pytest.raises(TypeError):
        config.set_default_float(np.float32)
    with pytest.raises(ValueError):
        config.set_default_positive_bijector("nope")
    with pytest.raises(ValueError):
        config.set_default_jitter(-1.0)


def test_model_construction_and_module_traversal():
    import gpflow_amd as gpflow
    k = gpflow.kernels.SquaredExponential(variance=2.0, lengthscales=[1.0, 3.0])
    assert k.ard and gpflow.kernels.RBF is gpflow.kernels.SquaredExponential
    with pytest.raises(TypeError, match="Unknown keyword argument"):
        gpflow.kernels.SquaredExponential(foo=1)
    with pytest.raises(ValueError, match="does not match size of ard parameter"):
        gpflow.kernels.RBF(lengthscales=[1.0, 2.0], active_dims=[0, 1, 2])
    lik = gpflow.likelihoods.Gaussian(0.5)
    assert lik.noise_variance() == pytest.approx(0.5)
    assert gpflow.likelihoods.Gaussian(scale=2.0).noise_variance() == pytest.approx(4.0)
    with pytest.raises(AssertionError):
        gpflow.likelihoods.Gaussian(1.0, scale=1.0)
    with pytest.raises(ValueError):
        gpflow.likelihoods.Gaussian(1e-7)  # below the 1e-6 lower bound
    Z = np.random.default_rng(0).normal(size=(4, 2))
    m = gpflow.models.SVGP(k, lik, Z, num_latent_gps=3)
    assert m.q_mu.shape == (4, 3) and m.q_sqrt.shape == (3, 4, 4)
    np.testing.assert_array_equal(m.q_sqrt.numpy()[1], np.eye(4))
    md = gpflow.models.SVGP(k, lik, Z, num_latent_gps=2, q_diag=True)
    assert md.q_sqrt.shape == (4, 2)
    m2 = gpflow.models.SVGP(k, lik, gpflow.inducing_variables.InducingPoints(Z), q_mu=np.zeros((4, 5)),
                            q_sqrt=np.stack([np.eye(4)] * 5))
    assert m2.num_latent_gps == 5
    names = set(gpflow.utilities.parameter_dict(m).keys())
    assert {".kernel.variance", ".kernel.lengthscales", ".likelihood.variance", ".inducing_variable.Z", ".q_mu",
            ".q_sqrt"} <= names
    assert len(m.trainable_parameters) == 6
    gpflow.set_trainable(m.kernel, False)
    assert len(m.trainable_parameters) == 4
    vals = gpflow.utilities.read_values(m)
    gpflow.utilities.multiple_assign(m, {".kernel.variance": 3.0})
    assert float(m.kernel.variance.numpy()) == pytest.approx(3.0) and vals[".kernel.variance"] == pytest.approx(2.0)
    cls = gpflow.posteriors.get_posterior_class
    iv = gpflow.inducing_variables.InducingPoints(Z)
    assert cls(k, iv) is gpflow.posteriors.IndependentPosteriorSingleOutput
    shared = gpflow.kernels.SharedIndependent(k, 3)
    siv = gpflow.inducing_variables.SharedIndependentInducingVariables(iv)
    assert cls(shared, siv) is gpflow.posteriors.IndependentPosteriorMultiOutput
    with pytest.raises(NotImplementedError):
        cls(shared, iv)
    assert gpflow.posteriors._validate_precompute_cache_type("Tensor") is gpflow.posteriors.PrecomputeCacheType.TENSOR
    assert gpflow.posteriors._validate_precompute_cache_type(None) is gpflow.posteriors.PrecomputeCacheType.NOCACHE
    with pytest.raises(ValueError):
        gpflow.posteriors._validate_precompute_cache_type(3)


def test_shard_bounds():
    from gpflow_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8192, 8193):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


# ---------------------------------------------------------------- host logic of the rows added for SURVEY 8f
def test_bijector_forward_grad_matches_finite_differences():
    from gpflow_amd.base import Chain, Exp, Identity, Shift, Softplus, positive
    x = np.array([-4.0, -0.3, 0.0, 0.7, 6.0])
    h = 1e-6
    for b in (Identity(), Softplus(), Exp(), Shift(0.5), Chain([Shift(1e-3), Softplus()]), positive(lower=1e-6),
              positive(base="exp")):
        fd = (b.forward(x + h) - b.forward(x - h)) / (2 * h)
        np.testing.assert_allclose(b.forward_grad(x), fd, rtol=1e-6, atol=1e-9, err_msg=b.name)
    from gpflow_amd.base import FillTriangular
    with pytest.raises(NotImplementedError):
        FillTriangular().forward_grad(np.zeros(3))


def test_gradient_entry_points_refuse_unsupported_models_before_touching_the_device():
    """The reverse pass covers the SVGP (whitened or not) / GPR / SGPR with ONE SquaredExponential or Matern kernel and a
    Gaussian likelihood; everything else must say so (NotImplementedError), not silently compute something else."""
    import gpflow_amd as gpflow
    from gpflow_amd import training
    Z = np.random.default_rng(0).normal(size=(5, 2))
    lik = gpflow.likelihoods.Gaussian(0.1)
    summed = gpflow.models.SVGP(gpflow.kernels.Matern32() + gpflow.kernels.SquaredExponential(), lik, Z)
    qdiag = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(), lik, Z, q_diag=True)
    sliced = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(active_dims=[0]), lik, Z)
    data = (np.zeros((4, 2)), np.zeros((4, 1)))
    with pytest.raises(NotImplementedError):      # natural gradients need the full q_sqrt (optimizers/natgrad.py)
        training.SVGPTrainer(qdiag, natgrad_gamma=0.1)
    for m in (summed, qdiag, sliced):
        with pytest.raises(NotImplementedError):
            gpflow.optimizers.NaturalGradient(1.0).minimize(m, data)
    # (round 3: SVGP.elbo_and_grad itself covers q_diag, active_dims and the Matern families; round 4: Sum / Product of
    #  stationary kernels; round 5: members with their own active_dims, combinations under the un-whitened SVGP and SGPR --
    #  a diagonal q_sqrt under a combination, a heteroskedastic likelihood under either SVGP -- tests/test_gpu_gradients.py.
    #  and under the device-resident trainer, nested combinations.  Still out in the reverse pass: combinations holding something other
    #  than SquaredExponential / Matern members)
    odd = gpflow.models.SVGP(gpflow.kernels.Matern32() + gpflow.kernels.SeparateIndependent([gpflow.kernels.SquaredExponential()]), lik, Z)
    with pytest.raises(NotImplementedError):
        odd.elbo_and_grad(data)
    with pytest.raises(NotImplementedError):
        gpflow.optimizers.Scipy().minimize(object())


def test_adam_state_matches_tf_keras_rule():
    from gpflow_amd.training import _Adam
    opt = _Adam(1e-2, 0.9, 0.999, 1e-7)
    p = np.array([1.0, -2.0]); m = np.zeros(2); v = np.zeros(2)
    for t, g in enumerate([np.array([0.3, -0.1]), np.array([0.2, 0.4]), np.array([-0.5, 0.1])], start=1):
        opt.t = t
        p_new = opt.update_host("p", p, g)
        m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
        lr_t = 1e-2 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        np.testing.assert_allclose(p_new, p - lr_t * m / (np.sqrt(v) + 1e-7), rtol=1e-14)
        p = p_new


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself and rank 0 prints ONE JSON line with n_gpus = 2
    (the launch / barrier / MAX-over-ranks protocol, exercised on CPU through --dry-launch: gloo, no device work)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "4", "--warmup", "1",
                        "--workload", "c4-strong"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["warmup"] == 1 and rec["scaling"] == "strong"
    assert rec["dry_launch"] is True and rec["allreduce_check"] is True


def test_bench_refuses_fewer_devices_than_ranks():
    """Without enough devices a multi-rank bench must fail loudly, not run a silent 1-GPU job."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("node has two devices")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0
    assert "needs 2 HIP devices" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_gaussian_rewraps_a_parameter_with_the_lower_bound_transform():
    """prepare_parameter_or_function (utilities/parameter_or_function.py:27-39) wraps ANY non-Function value -- a Parameter too --
    in Parameter(value, transform=positive(lower_bound)): a variance handed in with an identity transform must not be able to
    leave the bound, and prior / trainable carry over (base.py:155-161)."""
    import numpy as np
    import gpflow_amd as gpflow
    from gpflow_amd.base import Parameter
    raw = Parameter(0.3, trainable=False)                      # identity transform
    lik = gpflow.likelihoods.Gaussian(variance=raw)
    assert lik.variance is not raw and lik.variance.trainable is False
    assert abs(float(lik.variance.numpy()) - 0.3) < 1e-15
    lik.variance.assign_unconstrained(np.array(-50.0))        # far below anything an optimiser would reach
    assert float(lik.variance.numpy()) > lik.variance_lower_bound * (1 - 1e-12)


def test_check_info_names_a_timed_out_handoff():
    """include/gpk.h "info": INT_MAX is a timed-out internal hand-off, not a pivot column."""
    import torch
    from gpflow_amd import _lib, ops
    with pytest.raises(_lib.GpkError, match="hand-off timed out"):
        ops.check_info(torch.tensor([2 ** 31 - 1], dtype=torch.int32))
    with pytest.raises(_lib.GpkError, match="non-positive pivot at column 4"):
        ops.check_info(torch.tensor([5], dtype=torch.int32))
