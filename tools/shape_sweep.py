"""Shape sweep of one SVGP ELBO shard (whitened, P = 1; plus P = 4 shared at M = 1024) over minibatch rows x inducing points:
ms per step, TFLOP/s and fraction of the fp64 peak, and every point that lies more than 10 % above the interpolation of its
neighbours (along rows and along M).  potrf_core picks between ~10 schedules on row-count / M thresholds that were tuned on
the bench shapes (8192 and 1024 rows, M in {1024, 2048}); this table is the evidence that there is no cliff between them.
   python tools/shape_sweep.py > profiles/r06_shape_sweep.txt
Reference path being timed: SVGP.elbo = conditionals/util.py:84-169 + kullback_leiblers.py:59-165 + svgp.py:166-181."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops  # noqa: E402

PEAK = 78.6
ROWS = [512, 1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384]
MS = [512, 1024, 1536, 2048, 3072, 4096]
D = 8


def flops(m, b, p):
    return m ** 3 / 3.0 + float(m) * m * b * (1 + p)


def time_shape(m, rows, P, steps=12, warm=3):
    rng = np.random.default_rng(1)
    Z = ops.to_device(rng.normal(size=(m, D))); Xb = ops.to_device(rng.normal(size=(rows, D)))
    Yb = ops.to_device(rng.normal(size=(rows, P)))
    q_mu = ops.to_device(0.1 * rng.normal(size=(m, P)))
    q_sqrt = ops.to_device(np.tril(0.05 * rng.normal(size=(P, m, m))) + 0.5 * np.eye(m))
    ls = np.sqrt(D) * (0.8 + 0.05 * np.arange(D))
    ws = ops.svgp_elbo_workspace(m, rows, D, P, False)
    kw = dict(variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, ws=ws)
    for _ in range(warm):
        out, info = ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(steps):
            out, info = ops.svgp_elbo_shard(Z, Xb, Yb, q_mu, q_sqrt, **kw)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    assert int(info.cpu()[0]) == 0
    return best


def report(P, ms_list, rows_list):
    t = {(m, r): time_shape(m, r, P) for m in ms_list for r in rows_list}
    print(f"# whitened, P = {P}{' (shared kernel)' if P > 1 else ''}, D = {D}: ms per step | TFLOP/s | fraction of {PEAK} TFLOP/s")
    print("rows \\ M " + "".join(f"{m:>22d}" for m in ms_list))
    for r in rows_list:
        cells = []
        for m in ms_list:
            tf = flops(m, r, P) / t[(m, r)] / 1e12
            cells.append(f"{t[(m, r)] * 1e3:8.3f} {tf:6.1f} {tf / PEAK:5.3f} ")
        print(f"{r:8d} " + "".join(f"{c:>22s}" for c in cells))
    flagged = []
    for i, r in enumerate(rows_list):
        for j, m in enumerate(ms_list):
            if 0 < i < len(rows_list) - 1:   # along rows: linear in the row count between the neighbours
                r0, r1 = rows_list[i - 1], rows_list[i + 1]
                lin = t[(m, r0)] + (t[(m, r1)] - t[(m, r0)]) * (r - r0) / (r1 - r0)
                if t[(m, r)] > 1.10 * lin:
                    flagged.append(f"M={m} rows={r}: {t[(m, r)] * 1e3:.3f} ms vs {lin * 1e3:.3f} interpolated along rows (+{(t[(m, r)] / lin - 1) * 100:.0f} %)")
            if 0 < j < len(ms_list) - 1:     # along M: the step is ~ a M^3 + b M^2 rows: interpolate log t against log M
                m0, m1 = ms_list[j - 1], ms_list[j + 1]
                w = (np.log(m) - np.log(m0)) / (np.log(m1) - np.log(m0))
                geo = np.exp((1 - w) * np.log(t[(m0, r)]) + w * np.log(t[(m1, r)]))
                if t[(m, r)] > 1.10 * geo:
                    flagged.append(f"M={m} rows={r}: {t[(m, r)] * 1e3:.3f} ms vs {geo * 1e3:.3f} interpolated along M (+{(t[(m, r)] / geo - 1) * 100:.0f} %)")
    print("# points more than 10 % above the interpolation of their neighbours: " + ("none" if not flagged else ""))
    for f in flagged:
        print("#   " + f)
    print()


if __name__ == "__main__":
    report(1, MS, ROWS)
    report(4, [1024], ROWS)
