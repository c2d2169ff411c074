"""CPU: the WHOLE host layer (models, posteriors, conditionals, KL, covariances, kernels, likelihoods) driven end to
end against the oracle with the device primitives replaced by their CPU emulation (tests/fake_ops.py, which
reproduces each kernel's read / write contract).  The test bodies are the GPU parity tests of tests/test_gpu_models.py
themselves -- imported here without that module's `gpu` mark and given an emulated `gp` fixture -- so the routing,
shapes, transposes, broadcasting and error behaviour of the Python mirror are exercised in the `-m "not gpu"` tier too.
What this does NOT test is the HIP kernels: that is what the same bodies do under `-m gpu`.
"""
import pytest

import fake_ops
import test_gpu_models as T


@pytest.fixture
def gp(monkeypatch):
    import gpflow_amd
    from gpflow_amd import ops
    for name in dir(fake_ops):
        if name.startswith("_") or not callable(getattr(fake_ops, name)) or not hasattr(ops, name):
            continue
        if name in ("torch", "np", "sla"):
            continue
        monkeypatch.setattr(ops, name, getattr(fake_ops, name))
    return gpflow_amd


@pytest.fixture
def gpu(gp):
    """The next-row GPU tests (gradients, trainer, SGPR, natural gradient) ask for the `gpu` fixture: under emulation it
    is the patched primitive layer and a CPU device."""
    import torch
    return torch.device("cpu")


_names = [n for n in dir(T) if n.startswith("test_")]
for _n in _names:
    globals()[_n] = getattr(T, _n)
import test_gpu_gradients as TG  # noqa: E402
import test_gpu_reference_golden as TR  # noqa: E402  (values computed by the reference's own source)
import test_gpu_sgpr as TS  # noqa: E402

for _mod, _pre in ((TG, "grad"), (TS, "sgpr"), (TR, "refgolden")):
    for _n in [n for n in dir(_mod) if n.startswith("test_")]:
        globals()[f"test_{_pre}_{_n[5:]}"] = getattr(_mod, _n)
del _n, _mod, _pre
