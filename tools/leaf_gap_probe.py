"""Begin time and gap of every leaf of one SVGP shard (M = argv[1], rows = argv[2]) from the experimental library's in-kernel
wall-clock stamps -- an UNPROFILED timeline of the latency chain (rocprofv3 triples the cost of a host call):
   GPK_LIBRARY=gpflow_amd/libgpk_exp.so GPK_LEAF_DBG=1 python tools/leaf_gap_probe.py 2048 1024"""
import ctypes, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import _lib, ops  # noqa: E402
lib = _lib.load()
dump = ctypes.CDLL(os.environ["GPK_LIBRARY"]).gpk_exp_leaf_dbg_dump
m, B, d = int(sys.argv[1]), int(sys.argv[2]), 8
rng = np.random.default_rng(0)
ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
Z = ops.to_device(rng.normal(size=(m, d))); Xb = ops.to_device(rng.normal(size=(B, d))); Yb = ops.to_device(rng.normal(size=(B, 1)))
q_mu = ops.to_device(0.1 * rng.normal(size=(m, 1)))
q_sqrt = ops.to_device((np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m))[None])
ws = ops.svgp_elbo_workspace(m, B, d, 1, False)
for _ in range(4):
    out, info = ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, ws=ws)
torch.cuda.synchronize()
print(f"== SVGP shard M={m} rows={B} (fourth call)", flush=True)
dump(3 * ((m + 127) // 128))
