"""Phase timers (load / factor / invert / store) of every leaf of ONE N = 16384 GPR factorisation and of one SVGP step, from
the experimental library's in-kernel wall-clock stamps:  GPK_LIBRARY=gpflow_amd/libgpk_exp.so GPK_LEAF_DBG=1 python tools/leaf_phase_probe.py"""
import ctypes
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dump = ctypes.CDLL(os.environ["GPK_LIBRARY"]).gpk_exp_leaf_dbg_dump
rng = np.random.default_rng(0)
n, d = 16384, 8
X = ops.to_device(rng.normal(size=(n, d)))
Y = ops.to_device(rng.normal(size=(n, 1)))
ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
for _ in range(2):
    out, info = ops.gpr_lml(X, Y, variance=1.0, lengthscales=ls, noise_variance=0.1)
torch.cuda.synchronize()
print("== GPR N=16384 (second call)", flush=True)
sys.stdout.flush()
dump(128)
m, B = 2048, 8192
Z = ops.to_device(rng.normal(size=(m, d)))
Xb = ops.to_device(rng.normal(size=(B, d)))
Yb = ops.to_device(rng.normal(size=(B, 1)))
q_mu = ops.to_device(0.1 * rng.normal(size=(m, 1)))
q_sqrt = ops.to_device((np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m))[None])
ws = ops.svgp_elbo_workspace(m, B, d, 1, False)
for _ in range(3):
    out, info = ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, ws=ws)
torch.cuda.synchronize()
print("== SVGP Cm step (third call)", flush=True)
dump(32)
