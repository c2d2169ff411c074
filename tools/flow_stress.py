"""Run-to-run bit-reproducibility of the fused SVGP shard at config Cm (the dataflow kernel's coherence check): N calls
on the same inputs must return identical bits, and the value must match the oracle."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for (m, B, d, P) in [(2048, 8192, 8, 1), (1152, 777, 4, 3), (1024, 2500, 8, 2)]:
    rng = np.random.default_rng(5)
    Z = ops.to_device(rng.normal(size=(m, d))); Xb = ops.to_device(rng.normal(size=(B, d)))
    Yb = ops.to_device(rng.normal(size=(B, P))); q_mu = ops.to_device(0.1 * rng.normal(size=(m, P)))
    q_sqrt = ops.to_device(np.stack([np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m) for _ in range(P)]))
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    ws = ops.svgp_elbo_workspace(m, B, d, P, False)
    outs = []
    for _ in range(reps):
        out, info = ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, ws=ws)
        outs.append(out.cpu().numpy().copy())
    outs = np.array(outs)
    n_diff = int((outs != outs[0]).any(axis=1).sum())
    bad += n_diff
    print(f"m={m} B={B} P={P}: {reps} runs, {n_diff} differ from the first; spread {np.ptp(outs[:,0]):.3e} on {outs[0,0]:.9e}; info {int(info.cpu()[0])}")
print("STRESS", "OK" if bad == 0 else "NONDETERMINISTIC")
