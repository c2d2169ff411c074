"""One-shot GPU probe: micro-benchmarks + timings of the main pieces (writes gpurun_out/probe.json)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = ops.device()
res = {"device": torch.cuda.get_device_name(0)}


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return float(np.median(ts)), float(np.min(ts))


sink = torch.zeros(8, dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream
for blocks in (256, 512, 1024, 2048):
    iters = 20000
    t, tmin = timeit(lambda: lib.gpk_bench_mfma_f64(s, blocks, iters, sink.data_ptr()), reps=3, warm=1)
    res[f"mfma_f64_tflops_blocks{blocks}"] = blocks * 4 * iters * 8 * 2048 / tmin / 1e12
buf = torch.empty(1 << 28, dtype=torch.float64, device=dev)  # 2 GiB
t, tmin = timeit(lambda: lib.gpk_bench_stream_store(s, buf.data_ptr(), buf.numel()))
res["stream_store_TBps"] = buf.numel() * 8 / tmin / 1e12
del buf

rng = np.random.default_rng(0)
for n in (4096, 8192, 16384):
    A = ops.to_device(rng.normal(size=(n, 512)))
    C = torch.zeros((n, n), dtype=torch.float64, device=dev)
    t, tmin = timeit(lambda: ops.gemm_nt(A, A, alpha=-1.0, beta=1.0, C=C), reps=3, warm=1)
    res[f"gemm_nt_{n}x{n}x512_tflops"] = 2.0 * n * n * 512 / tmin / 1e12
    t, tmin = timeit(lambda: ops.gemm_nt(A, A, alpha=-1.0, beta=1.0, C=C, c_lower=True), reps=3, warm=1)
    res[f"syrk_lower_{n}x512_tflops"] = 1.0 * n * n * 512 / tmin / 1e12
    del A, C

for n, d in ((16384, 8),):
    X = ops.to_device(rng.normal(size=(n, d)))
    K = torch.empty((n, n), dtype=torch.float64, device=dev)
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    t, tmin = timeit(lambda: ops.kernel_matrix(X, None, variance=1.0, lengthscales=ls, diag_add=0.1, out=K))
    res[f"rbf_full_{n}_ms"] = tmin * 1e3
    res[f"rbf_full_{n}_TBps"] = n * n * 8 / tmin / 1e12
    t, tmin = timeit(lambda: ops.kernel_matrix(X, None, variance=1.0, lengthscales=ls, diag_add=0.1, lower_only=True, out=K))
    res[f"rbf_lower_{n}_ms"] = tmin * 1e3
    invd = ops.invd_alloc(n)

    def chol():
        ops.kernel_matrix(X, None, variance=1.0, lengthscales=ls, diag_add=0.1, lower_only=True, out=K)
        ops.potrf_(K, n, invd=invd)
    t, tmin = timeit(chol, reps=3, warm=1)
    res[f"rbf+potrf_{n}_ms"] = tmin * 1e3
    res[f"potrf_{n}_tflops_incl_rbf"] = n ** 3 / 3 / tmin / 1e12
    del K, X

for m in (1024, 2048):
    Z = ops.to_device(rng.normal(size=(m, 8)))
    K = torch.empty((m, m), dtype=torch.float64, device=dev)
    invd = ops.invd_alloc(m)
    ls = np.sqrt(8) * (0.8 + 0.05 * np.arange(8))

    def chol():
        ops.kernel_matrix(Z, None, variance=1.0, lengthscales=ls, diag_add=1e-2, lower_only=True, out=K)
        ops.potrf_(K, m, invd=invd)
    t, tmin = timeit(chol, reps=5, warm=2)
    res[f"potrf_{m}_ms"] = tmin * 1e3
    B = 8192
    Xb = ops.to_device(rng.normal(size=(B, 8)))
    Yb = ops.to_device(rng.normal(size=(B, 1)))
    q_mu = ops.to_device(0.1 * rng.normal(size=(m, 1)))
    q_sqrt = ops.to_device((np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m))[None])
    ws = ops.svgp_elbo_workspace(m, B, 8, 1, False)
    out = torch.empty(2, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    fn = lambda: ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1,
                                     jitter=1e-6, ws=ws, out=out, info=info)
    t, tmin = timeit(fn, reps=5, warm=2)
    res[f"svgp_step_M{m}_B{B}_ms"] = tmin * 1e3
    res[f"svgp_step_M{m}_B{B}_tflops"] = (m ** 3 / 3 + 2.0 * m * m * B) / tmin / 1e12

os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/probe.json", "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
