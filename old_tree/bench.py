"""bench.py -- headline benchmark of the dense-GP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cm|c3|c4-weak|c4-strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
`python bench.py --gpus N` WITHOUT a launcher spawns the N ranks itself (torch.distributed.run, 127.0.0.1, a free port) and
refuses to run when the node has fewer than N devices; `--dry-launch` exercises that launch / timing protocol on CPU (gloo).

Metric (BASELINE.json): SVGP ELBO steps/s at N=1e6, M=2048, D=8 (config "Cm", the default workload), with the GPR
config C2 (N=16384: K build + Cholesky + predict, GF/s vs fp64 peak) reported alongside in the same JSON line (key
"gpr_cholesky", N=1 only).

A "step" = one forward minibatch ELBO evaluation (SVGP.elbo, gpflow/models/svgp.py:166-181): Kuu / Kuf builds,
Cholesky of Kuu, the triangular solves, the q_sqrt projection, the variational expectations, KL, the (multi-GPU)
all-reduce of the per-shard data term and the scalar landing in host memory.  Inputs (the data matrix, Z, q) are
resident in HBM before the timed region.

Workloads:
  cm         N=1e6, M=2048, D=8,  8192 rows per GPU (weak scaling: a global step covers 8192*G rows)     [headline]
  c3         N=1e6, M=1024, D=8,  8192 rows per GPU (weak)
  c4-weak    N=1e7, M=2048, D=16, 8192 rows per GPU (weak)
  c4-strong  N=1e7, M=2048, D=16, global minibatch 8192 rows, 8192/G rows per GPU (strong scaling)
`value` = 8192-row minibatch evaluations per second over the whole job for the weak workloads, global steps per second
for c4-strong.  The last step's ELBO is checked against the CPU oracle ON THE SAME ARRAYS (`parity_rel_err`), and that
oracle call is what `cpu_baseline` times.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues per priority level (default 4).  The factorisation runs a
# latency-critical panel stream next to bulk streams; with 2 hardware queues the SVGP step measured faster than with 4
# (same-box A/B, tools/ab.sh: 2.6 vs 3.6 ms in round 2, 415 vs 380 steps/s in round 1).  Must be set before the HIP
# runtime initialises; an explicit setting in the environment wins.  The same value is used for every rank of a
# multi-GPU run (RCCL's stream then shares one of the two queues; it could not be measured on the 1-GPU boxes of this
# build -- set GPU_MAX_HW_QUEUES=4 in the environment if an 8-GPU node shows contention).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet, FP64 matrix (= FP64 vector); the in-image guide lists no fp64 row
WORKLOADS = {
    # name: (N, M, D, rows per global step, strong?, data seed)
    "cm": (1_000_000, 2048, 8, 8192, False, 4),
    "c3": (1_000_000, 1024, 8, 8192, False, 4),
    "c4-weak": (10_000_000, 2048, 16, 8192, False, 6),
    "c4-strong": (10_000_000, 2048, 16, 8192, True, 6),
}
P_LAT = 1


def host_threads() -> int:
    """Threads the CPU baseline actually runs on: the BLAS pool behind NumPy / SciPy (one definition for every
    `cpu_baseline.cores`)."""
    try:
        import threadpoolctl
        return int(max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [os.cpu_count() or 1]))
    except Exception:
        return int(os.cpu_count() or 1)


def cpu_description() -> dict:
    """What SURVEY 8d asks to be stated next to a CPU number: CPU model, logical cores, the BLAS behind NumPy / SciPy and behind
    torch, and the thread environment."""
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    blas = []
    try:
        import threadpoolctl
        blas = [f"{p_.get('internal_api', '?')} {p_.get('version', '?')} x{p_.get('num_threads', '?')} ({p_.get('user_api', '?')})"
                for p_ in threadpoolctl.threadpool_info()]
    except Exception:
        pass
    env = {k: os.environ[k] for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS") if k in os.environ}
    return {"cpu_model": model, "nproc": int(os.cpu_count() or 1), "blas": blas, "thread_env": env or "unset (library defaults)",
            "torch_threads": int(torch.get_num_threads())}


def timed_median(fn, budget_s: float, max_reps: int = 5):
    """benchmark/run.py:71-122 convention: one warm-up call, then the median of the repetitions that fit the budget (>= 1)."""
    out = fn()
    times, t_start = [], time.perf_counter()
    while len(times) < max_reps and (not times or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return out, float(np.median(times)), len(times)


def svgp_step_flops(m: int, b: int, p: int) -> float:
    """Algorithmic flops of one whitened step: M^3/3 + M^2 B (1 + P)."""
    return m ** 3 / 3.0 + float(m) * m * b * (1 + p)


def make_inputs(n_data, m_ind, d_in, seed, device):
    """X ~ N(0,1), Y = sin(sum x) + 0.1 eps, Z = first M rows + 0.01 noise, q_mu ~ 0.1 N(0,1),
    q_sqrt = tril(0.05 N(0,1)) + 0.5 I, ARD lengthscales sqrt(D)(0.8 + 0.05 d), noise 0.1.  The rows are i.i.d., so
    minibatch s of rank r is simply a contiguous slice (a fixed permutation of i.i.d. rows changes nothing).
    Generated on the host from NumPy seeds (bit-reproducible anywhere; identical on every rank), then moved to HBM before
    the timed region."""
    rng = np.random.default_rng(seed)
    Xh = rng.standard_normal((n_data, d_in))
    Yh = np.sin(Xh.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((n_data, P_LAT))
    Zh = Xh[:m_ind] + 0.01 * rng.standard_normal((m_ind, d_in))
    q_mu_h = 0.1 * rng.standard_normal((m_ind, P_LAT))
    q_sqrt_h = np.tril(0.05 * rng.standard_normal((P_LAT, m_ind, m_ind))) + 0.5 * np.eye(m_ind)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    X, Y, Z, q_mu, q_sqrt = to(Xh), to(Yh), to(Zh), to(q_mu_h), to(q_sqrt_h)
    ls = np.sqrt(d_in) * (0.8 + 0.05 * np.arange(d_in))
    return X, Y, Z, q_mu, q_sqrt, ls


def cpu_baseline_and_parity(Xb, Yb, Z, q_mu, q_sqrt, ls, n_data, gpu_elbo, budget_s: float = 14.0):
    """The CPU column, on EXACTLY the arrays of the last timed step, two implementations of the reference's algorithm
    (GPflow + TensorFlow themselves are not installable here):
      * the NumPy/SciPy oracle -- its value is the parity check of `last_elbo`; SciPy's solve_triangular is LAPACK dtrtrs, which
        OpenBLAS runs on one thread, so this leg alone would be a strawman;
      * a torch-CPU fp64 port with the reference's structure -- torch.linalg.cholesky, solve_triangular, the dense batched
        L_q^T A matmul (conditionals/util.py:67,125,151-157) -- on torch's intra-op pool: the "second opinion".
    `value` is the FASTER of the two (warm-up 1 call + median, benchmark/run.py:71-122 convention)."""
    from oracle import gp_oracle as orc
    from oracle import gp_oracle_grad as orct
    kw = dict(variance=1.0, lengthscales=ls, noise_variance=0.1, whiten=True, num_data=n_data)
    ref, med, nrep = timed_median(lambda: orc.svgp_elbo(Xb, Yb, Z, q_mu, q_sqrt, **kw), budget_s)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))  # noqa: E731
    Xt, Yt, Zt, qmt, qst, lst = t(Xb), t(Yb), t(Z), t(q_mu), t(q_sqrt), t(np.atleast_1d(ls))
    with torch.no_grad():
        ref_t, med_t, nrep_t = timed_median(
            lambda: float(orct.svgp_elbo_torch(Xt, Yt, Zt, qmt, qst, torch.tensor(1.0, dtype=torch.float64), lst,
                                               torch.tensor(0.1, dtype=torch.float64), num_data=n_data, whiten=True)), budget_s)
    m, b, d = Z.shape[0], Xb.shape[0], Xb.shape[1]
    what = f"ELBO steps on the arrays of the last timed GPU step (M={m}, B={b}, D={d}, P=1, whitened)"
    legs = {"numpy_oracle": {"value": 1.0 / med, "unit": "steps/s", "threads": int(host_threads()), "reps": nrep,
                             "impl": "NumPy/SciPy (OpenBLAS): dpotrf + dtrtrs (single-threaded in OpenBLAS) + dgemm"},
            "second_opinion": {"value": 1.0 / med_t, "unit": "steps/s", "threads": int(torch.get_num_threads()), "reps": nrep_t,
                               "impl": "torch-CPU fp64: linalg.cholesky + solve_triangular + dense L_q^T A matmul (the reference's structure)",
                               "rel_err_vs_numpy_oracle": abs(ref_t - ref) / abs(ref)}}
    best = "second_opinion" if legs["second_opinion"]["value"] >= legs["numpy_oracle"]["value"] else "numpy_oracle"
    base = {"value": legs[best]["value"], "unit": "steps/s", "cores": legs[best]["threads"], "kind": "port",
            "sample": f"{legs[best]['reps']} {what}, median {1e3 / legs[best]['value']:.1f} ms/step, the faster of two CPU ports "
                      f"({best}: {legs[best]['impl']}); GPflow+TensorFlow is not installable in this image",
            **legs, **cpu_description()}
    return base, float(ref), abs(gpu_elbo - ref) / abs(ref)


def train_step_leg(X, Y, Z, q_mu, q_sqrt, ls, n_data, b_rows, steps: int = 20):
    """SURVEY 8f row 1 (the caller of the hot path): one TRAINING step = forward + hand-written reverse pass
    (gpflow_amd/gradients.py) + Adam update, reported beside the headline ELBO metric (not part of it)."""
    from gpflow_amd import gradients, ops
    m_ind = Z.shape[0]
    kw = dict(variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, scale=float(n_data) / b_rows)
    n_batches = n_data // b_rows
    m = {k: torch.zeros_like(v) for k, v in (("Z", Z), ("q_mu", q_mu), ("q_sqrt", q_sqrt))}
    v2 = {k: torch.zeros_like(v) for k, v in m.items()}
    par = {"Z": Z.clone(), "q_mu": q_mu.clone(), "q_sqrt": q_sqrt.clone()}

    def one(s):
        lo = (s % n_batches) * b_rows
        F, g, info = gradients.svgp_elbo_and_grad(par["Z"], X[lo:lo + b_rows], Y[lo:lo + b_rows], par["q_mu"],
                                                  par["q_sqrt"], **kw)
        for k in par:  # Adam (tf.keras defaults; step size without bias correction as before) on the device-resident variables
            ops.adam_step_(par[k], g[k], m[k], v2[k], beta1=0.9, beta2=0.999, epsilon=1e-7, step=1e-3, maximise=True)
        small = torch.cat([g["variance"], g["lengthscales"], g["noise_variance"], F]).cpu()  # scalar grads + ELBO to host
        return float(small[-1]), int(info.cpu()[0])

    for s in range(3):
        one(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        elbo, info = one(3 + s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # counted, not guessed: HIP events + algorithmic flops of every GEMM launch of one more evaluation
    from gpflow_amd import _lib
    lib = _lib.load()
    lib.gpk_profile_gemm_enable(1)
    one(3 + steps)
    ms_g, n_g, fl_g = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
    lib.gpk_profile_gemm_collect_min(ctypes.c_double(0.0), 0, ctypes.byref(ms_g), ctypes.byref(n_g), ctypes.byref(fl_g))
    lib.gpk_profile_gemm_enable(0)
    flops = fl_g.value
    return {"workload": "SVGP training step (ELBO + gradients w.r.t. Z, q_mu, q_sqrt, kernel and noise parameters + Adam), "
                        f"M={m_ind}, {b_rows} rows", "ms_per_step": dt * 1e3, "steps_per_s": 1.0 / dt, "last_elbo": elbo,
            "info": info, "gemm_gflop_per_step_counted": flops / 1e9, "gemm_launches_per_step": int(n_g.value),
            "tflops": flops / dt / 1e12, "frac_of_fp64_peak": flops / dt / 1e12 / FP64_PEAK_TFLOPS,
            "note": "hyper-parameters held fixed in this leg (their gradients are computed and read back); "
                    "SVGPTrainer updates them on the host"}


def other_workloads_leg(device, with_oracle: bool, steps: int = 20):
    """Every other BASELINE config in the same driver-run line: C3 (M=1024), the C4 rank shard (D=16; 8192 rows
    = weak scaling, 1024 rows = one of 8 ranks of the strong-scaled step), C5 through the model surface with a shared and
    with separate kernels.  Inputs from NumPy seeds (a pool of 8 minibatches); ms/step = wall-clock of `steps` evaluations
    incl. the scalar landing on the host; parity of the LAST step against the oracle on the same arrays."""
    import gpflow_amd as gpflow
    from gpflow_amd import ops
    from oracle import gp_oracle as orc
    out = {}

    def pool(seed, b, d, p, m):
        rng = np.random.default_rng(seed)
        Xh = rng.normal(size=(8 * b, d))
        Yh = np.sin(Xh.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(8 * b, p))
        Z = Xh[:m] + 0.01 * rng.normal(size=(m, d))
        q_mu = 0.1 * rng.normal(size=(m, p))
        q_sqrt = np.stack([np.tril(0.05 * rng.normal(size=(m, m))) + 0.5 * np.eye(m) for _ in range(p)])
        return Xh, Yh, Z, q_mu, q_sqrt, np.sqrt(d) * (0.8 + 0.05 * np.arange(d))

    def run(name, note, step, flops, oracle_fn):
        for s in range(3):
            step(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            v = step(3 + s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        rec = {"workload": note, "ms_per_step": dt * 1e3, "steps_per_s": 1.0 / dt, "algorithmic_gflop_per_step": flops / 1e9,
               "step_tflops": flops / dt / 1e12, "step_frac_of_fp64_peak": flops / dt / 1e12 / FP64_PEAK_TFLOPS, "last_elbo": v}
        if with_oracle:
            t1 = time.perf_counter()
            ref = float(oracle_fn((3 + steps - 1) % 8))
            rec.update(oracle_elbo=ref, parity_rel_err=abs(v - ref) / abs(ref), oracle_seconds=time.perf_counter() - t1)
        out[name] = rec

    # C3 and the C4 shards through the fused C-ABI driver, exactly like the headline step
    for name, note, (n_data, m, d, b, seed) in (
            ("c3", "BASELINE config C3: SVGP N=1e6 M=1024 D=8, 8192 rows", (1_000_000, 1024, 8, 8192, 14)),
            ("c4_shard_8192", "BASELINE config C4 per-rank problem, weak scaling: N=1e7 M=2048 D=16, 8192 rows", (10_000_000, 2048, 16, 8192, 16)),
            ("c4_shard_1024", "BASELINE config C4 per-rank problem, one of 8 ranks of a strong-scaled 8192-row step: 1024 rows",
             (10_000_000, 2048, 16, 1024, 17))):
        Xh, Yh, Zh, qmh, qsh, ls = pool(seed, b, d, 1, m)
        Xd, Yd, Z, q_mu, q_sqrt = (ops.to_device(a) for a in (Xh, Yh, Zh, qmh, qsh))
        ws = ops.svgp_elbo_workspace(m, b, d, 1, False)
        o = torch.empty(2, dtype=torch.float64, device=device)
        inf = torch.zeros(1, dtype=torch.int32, device=device)
        scale = float(n_data) / b

        box = ops.HostMailbox(2)

        def step(s, Xd=Xd, Yd=Yd, Z=Z, q_mu=q_mu, q_sqrt=q_sqrt, ls=ls, ws=ws, o=o, inf=inf, b=b, scale=scale, box=box):
            lo = (s % 8) * b
            ops.svgp_elbo_shard(Z, Xd[lo:lo + b], Yd[lo:lo + b], q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1,
                                jitter=1e-6, ws=ws, out=o, info=inf)
            box.post(o, inf)  # scalars + status land in mapped host memory (gpk_publish_host), like the headline step
            h, status = box.wait()
            assert status == 0, status
            return float(h[0]) * scale - float(h[1])

        def orc_fn(i, Xh=Xh, Yh=Yh, Zh=Zh, qmh=qmh, qsh=qsh, ls=ls, b=b, n_data=n_data):
            return orc.svgp_elbo(Xh[i * b:(i + 1) * b], Yh[i * b:(i + 1) * b], Zh, qmh, qsh, variance=1.0, lengthscales=ls,
                                 noise_variance=0.1, whiten=True, num_data=n_data)
        run(name, note, step, svgp_step_flops(m, b, 1), orc_fn)
        del Xd, Yd, ws
    # Cm variants ("also report whiten=False, q_diag=True"): the headline shape through gpflow_amd.models.SVGP.elbo --
    # the un-whitened model is composed on the host from the primitives (one more triangular solve + the KL against Kuu),
    # the diagonal q goes through the fused driver with its row-statistics epilogue instead of the projection GEMM
    m, b, d = 2048, 8192, 8
    Xh, Yh, Zh, qmh, qsh, ls = pool(19, b, d, 1, m)
    Xd, Yd = ops.to_device(Xh), ops.to_device(Yh)
    qdh = 0.5 + 0.05 * np.abs(np.random.default_rng(20).normal(size=(m, 1)))
    for name, note, kwm, qs, okw, fl in (
            ("cm_unwhitened", "Cm shape, whiten=False (full q_sqrt): SVGP.elbo through the model surface -> gpk_svgp_elbo_shard(whiten=0): "
             "KL against N(0, Kuu) and the conditional on ONE factorisation, no second solve of the minibatch rows "
             "(algorithmic flops of THIS form: M^3/3 + 2 M^2 B + M^3/3 for the M rows of tril(q_sqrt)^T)",
             dict(whiten=False), qsh, dict(whiten=False), m ** 3 / 3.0 + 2.0 * float(m) * m * b + m ** 3 / 3.0),
            ("cm_q_diag", "Cm shape, q_diag=True (whitened): SVGP.elbo through the model surface (fused driver, no projection GEMM)",
             dict(whiten=True, q_diag=True), qdh, dict(whiten=True), m ** 3 / 3.0 + float(m) * m * b)):
        mv = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=ls), gpflow.likelihoods.Gaussian(0.1), Zh,
                                q_mu=qmh, q_sqrt=qs, num_data=1_000_000, **kwm)
        run(name, note, lambda s, mv=mv: float(mv.elbo((Xd[(s % 8) * b:(s % 8 + 1) * b], Yd[(s % 8) * b:(s % 8 + 1) * b]))), fl,
            lambda i, qs=qs, okw=okw: orc.svgp_elbo(Xh[i * b:(i + 1) * b], Yh[i * b:(i + 1) * b], Zh, qmh, qs, variance=1.0, lengthscales=ls,
                                                    noise_variance=0.1, num_data=1_000_000, **okw))
    del Xd, Yd
    # C5: multi-output SVGP, 4 latent GPs, M = 1024, through gpflow_amd.models.SVGP
    m, b, d, p = 1024, 8192, 8, 4
    Xh, Yh, Zh, qmh, qsh, ls = pool(18, b, d, p, m)
    Xd, Yd = ops.to_device(Xh), ops.to_device(Yh)
    shared_iv = gpflow.inducing_variables.SharedIndependentInducingVariables(gpflow.inducing_variables.InducingPoints(Zh))
    ms = gpflow.models.SVGP(gpflow.kernels.SharedIndependent(gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=ls), output_dim=p),
                            gpflow.likelihoods.Gaussian(0.1), shared_iv, q_mu=qmh, q_sqrt=qsh, num_latent_gps=p, num_data=1_000_000)
    run("c5_shared", "BASELINE config C5, SharedIndependent kernel + shared inducing points (one Cholesky, P-batched projection), "
        "4 latent GPs, M=1024, 8192 rows, through gpflow_amd.models.SVGP.elbo",
        lambda s: float(ms.elbo((Xd[(s % 8) * b:(s % 8 + 1) * b], Yd[(s % 8) * b:(s % 8 + 1) * b]))),
        m ** 3 / 3.0 + float(m) * m * b * (1 + p),
        lambda i: orc.svgp_elbo(Xh[i * b:(i + 1) * b], Yh[i * b:(i + 1) * b], Zh, qmh, qsh, variance=1.0, lengthscales=ls,
                                noise_variance=0.1, whiten=True, num_data=1_000_000))
    sep_var, sep_ls = [1.0, 0.8, 1.2, 0.9], [2.4, 2.8, 3.2, 3.6]
    ksep = gpflow.kernels.SeparateIndependent([gpflow.kernels.SquaredExponential(variance=v, lengthscales=l) for v, l in zip(sep_var, sep_ls)])
    shared_iv2 = gpflow.inducing_variables.SharedIndependentInducingVariables(gpflow.inducing_variables.InducingPoints(Zh))
    mp = gpflow.models.SVGP(ksep, gpflow.likelihoods.Gaussian(0.1), shared_iv2, q_mu=qmh, q_sqrt=qsh, num_latent_gps=p, num_data=1_000_000)
    run("c5_separate", "BASELINE config C5, SeparateIndependent kernels (batched [4,M,M] Cholesky + batched solves), 4 latent GPs, "
        "M=1024, 8192 rows, whitened, through gpflow_amd.models.SVGP.elbo",
        lambda s: float(mp.elbo((Xd[(s % 8) * b:(s % 8 + 1) * b], Yd[(s % 8) * b:(s % 8 + 1) * b]))),
        p * (m ** 3 / 3.0 + 2.0 * float(m) * m * b),
        lambda i: orc.svgp_elbo_separate(Xh[i * b:(i + 1) * b], Yh[i * b:(i + 1) * b], [Zh] * p, qmh, qsh, variances=sep_var,
                                         lengthscales_list=sep_ls, noise_variance=0.1, whiten=True, num_data=1_000_000))
    return out


def _event_time(fn, reps: int, warm: int = 1):
    """median device time of fn() over reps, by HIP events on the current stream (the caller's stream: the library
    forks from / joins to it, so the interval covers all internal streams)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return float(np.median(ts)), float(np.min(ts))


def gpr_leg(ops, lib, device, with_oracle: bool):  # noqa: C901
    """GPR config C2: N=16384, D=8, ARD lengthscales, noise 0.1:
    (i) K(X,X)+noise build + Cholesky + LML tail, one gpk_gpr_lml call (gpr.py:91-107);
    (ii) predict_f at T=4096 fresh rows, fused route (posteriors.py:435-443): factorisation with the test rows and
         (Y-m)^T riding along + reductions (the alpha-form of SURVEY 8d: N^3/3 + N^2 T flops)."""
    import gpflow_amd as gpflow
    n, d, T = 16384, 8, 4096
    rng = np.random.default_rng(2)
    Xh = rng.normal(size=(n, d))
    Yh = np.sin(Xh.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(n, 1))
    Xnew = ops.to_device(np.random.default_rng(3).normal(size=(T, d)))
    X, Y = ops.to_device(Xh), ops.to_device(Yh)
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    ws = torch.empty(int(lib.gpk_gpr_lml_workspace_bytes(n, d, 1)) // 8 + 1, dtype=torch.float64, device=device)
    kw = dict(variance=1.0, lengthscales=ls, noise_variance=0.1, ws=ws)
    res = {}

    def lml_call():
        res["out"], res["info"] = ops.gpr_lml(X, Y, **kw)
    t, _ = _event_time(lml_call, 5, warm=2)
    out, info = res["out"], res["info"]
    K = torch.empty((n, n), dtype=torch.float64, device=device)
    _, tkb = _event_time(lambda: ops.kernel_matrix(X, None, variance=1.0, lengthscales=ls, diag_add=0.1, out=K), 4)
    del K
    flops = n ** 3 / 3.0
    kb_alg = n * n * 8 + n * d * 8  # algorithmic bytes: the full N x N fp64 write + the N x D read
    # the trailing update on its own (north_star: ">= 60 % of fp64 MFMA peak on the N=16384 Cholesky trailing update"):
    # HIP events around every GEMM launch of one more factorisation; the outer rest-updates  A22 -= P P^T  (lower tiles,
    # K = 640) are the launches with >= 2e10 algorithmic flop (look-ahead strips and panel-internal GEMMs are smaller)
    lib.gpk_profile_gemm_enable(1)
    ops.gpr_lml(X, Y, **kw)
    ms_t, n_t, fl_t = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
    lib.gpk_profile_gemm_collect_min(ctypes.c_double(2e10), 1, ctypes.byref(ms_t), ctypes.byref(n_t), ctypes.byref(fl_t))
    win_ms, win_all, win_match, win_n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
    lib.gpk_profile_gemm_window(ctypes.c_double(2e10), ctypes.byref(win_ms), ctypes.byref(win_all), ctypes.byref(win_match),
                                ctypes.byref(win_n))
    ms_g, n_g, fl_g = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
    lib.gpk_profile_gemm_collect_min(ctypes.c_double(0.0), 0, ctypes.byref(ms_g), ctypes.byref(n_g), ctypes.byref(fl_g))
    lib.gpk_profile_gemm_enable(0)
    tu_tf = fl_t.value / (ms_t.value * 1e-3) / 1e12 if ms_t.value > 0 else 0.0
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pj = json.load(f)
        kb_traffic = float(pj["rbf_kernel<0> full 16384^2"]["hbm_bytes_per_launch"])
        traffic = float(pj["gemm_nt_fast<0,false> trailing update"]["hbm_bytes_per_launch"])
        predict_traffic = float(pj["GPR predict_f fused, whole call"]["hbm_bytes_per_call"]) \
            if "GPR predict_f fused, whole call" in pj else None
    except Exception:
        kb_traffic = None
        predict_traffic = None
    trailing = {"bound": "mfma", "kernel": "gemm_nt_fast<0,false>, lower tiles, K = 640 (outer trailing updates)",
                "achieved": tu_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tu_tf / FP64_PEAK_TFLOPS,
                "traffic": traffic,
                "launches": int(n_t.value), "algorithmic_gflop": fl_t.value / 1e9,
                "share_of_factorisation_flops": fl_t.value / flops, "summed_launch_ms": ms_t.value,
                "all_gemm_launches": int(n_g.value), "all_gemm_gflop": fl_g.value / 1e9,
                "phase_chipwide": {
                    "window_ms": win_ms.value, "gemm_launches_in_window": int(win_n.value),
                    "algorithmic_gflop_in_window": win_all.value / 1e9,
                    "achieved": win_all.value / (win_ms.value * 1e-3) / 1e12 if win_ms.value > 0 else 0.0,
                    "frac": (win_all.value / (win_ms.value * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if win_ms.value > 0 else 0.0,
                    "unit": "TFLOP/s",
                    "note": "first start .. last end of those launches (HIP events) and the algorithmic flops of EVERY GEMM "
                            "issued in between -- the trailing updates plus the look-ahead panel (solves, inner updates, "
                            "strips) that shares the chip with them; leaf kernels not counted"},
                "note": "sum of algorithmic flops / sum of HIP-event durations of these launches, recorded on the bulk "
                        "stream (CU-masked: 224 of 256 CUs, persistent workgroups fed from a tile queue; the look-ahead panel "
                        "runs beside them on the other 32), divided by the WHOLE chip's peak"}
    # (ii) predict_f at T = 4096 through the model surface (fused route), then the cached posterior
    m = gpflow.models.GPR((X, Y), gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=ls), noise_variance=0.1)
    pred = {}

    def fused():
        pred["mu"], pred["var"] = m.predict_f(Xnew)
    t_pred, _ = _event_time(fused, 3)
    post = m.posterior()
    torch.cuda.synchronize()
    cached = {}

    def cached_call():
        cached["mu"], cached["var"] = post.predict_f(Xnew)
    t_cached, _ = _event_time(cached_call, 3)
    pred_flops = n ** 3 / 3.0 + float(n) * n * T
    predict = {"workload": f"GPR.predict_f(Xnew [{T},{d}]) fused: K build + Cholesky with the test rows riding along + "
                           "reductions (alpha-form mean, SURVEY 8d)",
               "ms_total": t_pred * 1e3, "algorithmic_gflop": pred_flops / 1e9,
               "roofline": {"bound": "mfma", "achieved": pred_flops / t_pred / 1e12, "peak": FP64_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": pred_flops / t_pred / 1e12 / FP64_PEAK_TFLOPS,
                            "kernel": "whole call (gemm_nt_fast launches carry > 97 % of the flops)", "traffic": predict_traffic,
                            "traffic_note": "FETCH_SIZE x2 + WRITE_SIZE summed over EVERY kernel of one fused call (PMC passes on "
                                            "tools/predict_pmc_probe.py, profiles/pmc_traffic.json); algorithmic bytes of the call: "
                                            "the N^2/2 lower triangle written and read once + the T x N test rows = "
                                            f"{(n * n / 2 * 8 * 2 + T * n * 8 * 2) / 1e9:.2f} GB"},
               "cached_posterior_ms": t_cached * 1e3, "cached_posterior_gflop": float(n) * n * T / 1e9,
               "cached_posterior_tflops": float(n) * n * T / t_cached / 1e12,
               "routes_max_abs_diff": {"mean": float((pred["mu"] - cached["mu"]).abs().max()),
                                       "var": float((pred["var"] - cached["var"]).abs().max())}}
    res = {"workload": "GPR RBF N=16384 D=8 fp64: K build + Cholesky + LML (one gpk_gpr_lml call) and predict_f at T=4096",
           "kernel_build_roofline": {"bound": "hbm", "kernel": "rbf_kernel (full N x N)", "achieved": kb_alg / tkb / 1e9,
                                     "peak": 8000.0, "unit": "GB/s", "frac": kb_alg / tkb / 8e12, "traffic": kb_traffic},
           "trailing_update_roofline": trailing, "predict": predict,
           "lml": float(out.cpu()[0]), "info": int(info.cpu()[0]), "ms_total": t * 1e3,
           "cholesky_gflops_incl_build_and_tail": flops / t / 1e9,
           "frac_of_fp64_peak": flops / t / 1e12 / FP64_PEAK_TFLOPS,
           "kernel_build_full_ms": tkb * 1e3, "kernel_build_full_GBps": n * n * 8 / tkb / 1e9,
           "kernel_build_frac_of_8TBps": n * n * 8 / tkb / 8e12}
    if with_oracle:
        from oracle import gp_oracle as orc
        t0 = time.perf_counter()
        ref = orc.gpr_log_marginal_likelihood(Xh, Yh, variance=1.0, lengthscales=ls, noise_variance=0.1)
        t_cpu = time.perf_counter() - t0
        res["parity_rel_err"] = abs(res["lml"] - ref) / abs(ref)
        res["oracle_lml"] = float(ref)
        # second opinion: the same computation on torch-CPU fp64 with the reference's structure (gpr.py:91-107)
        from oracle import gp_oracle_grad as orct
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))  # noqa: E731
        with torch.no_grad():
            t1 = time.perf_counter()
            ref_t = float(orct.gpr_lml_torch(tt(Xh), tt(Yh), torch.tensor(1.0, dtype=torch.float64), tt(np.atleast_1d(ls)),
                                             torch.tensor(0.1, dtype=torch.float64)))
            t_torch = time.perf_counter() - t1
        legs = {"numpy_oracle": {"value": flops / t_cpu / 1e9, "unit": "GF/s", "threads": host_threads(), "seconds": t_cpu,
                                 "impl": "NumPy/SciPy (OpenBLAS): K build + dpotrf + dtrtrs"},
                "second_opinion": {"value": flops / t_torch / 1e9, "unit": "GF/s", "threads": int(torch.get_num_threads()),
                                   "seconds": t_torch, "impl": "torch-CPU fp64: K build + linalg.cholesky + solve_triangular",
                                   "rel_err_vs_numpy_oracle": abs(ref_t - ref) / abs(ref)}}
        best = "second_opinion" if t_torch <= t_cpu else "numpy_oracle"
        res["cpu_baseline"] = {"value": legs[best]["value"], "unit": "GF/s", "cores": legs[best]["threads"], "kind": "port",
                               "sample": f"one GPR.log_marginal_likelihood at N={n} on the same arrays, the faster of two CPU ports "
                                         f"({best}, {legs[best]['seconds']:.1f} s: {legs[best]['impl']}); N^3/3 flops counted",
                               **legs, **cpu_description()}
    return res


def stream_selfcheck(lib):
    """The library's init-time check of its stream -> hardware-queue layout (gpk_stream_selfcheck): a slow process is detected
    (and repaired) instead of benchmarked."""
    now, first, rec = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), ctypes.c_int()
    if lib.gpk_stream_selfcheck(now, first, ctypes.byref(rec)) != 0:
        return None
    mode = int(lib.gpk_chain_handoff_mode())
    return {"handoff_us_P_X_Bs_pairs": [round(v, 1) for v in now], "first_layout_us": [round(v, 1) for v in first],
            "streams_recreated": bool(rec.value), "limit_us": 30,
            "chain_handoff": {2: "flag words written / awaited by kernels only: entry signals, gate + store kernels, bounded in-kernel polls "
                                 "(no queue packets on the chain, no stream memory operations)",
                              1: "stream memory operations + in-kernel polls (no event packets on the chain)",
                              0: "events (kernels of two streams were NOT seen running concurrently: a serialising tool is attached)"
                              }.get(mode, "not initialised")}


def spawn_ranks(n: int, dry: bool) -> int:
    """Re-execute this script under torch.distributed.run with n ranks on this node (what the driver's command line does);
    returns the launcher's exit status.  Refuses -- loudly, before starting anything -- when the node has fewer devices."""
    import socket
    import subprocess
    if not dry and torch.cuda.device_count() < n:
        print(f"bench.py: --gpus {n} needs {n} HIP devices, this node has {torch.cuda.device_count()}; not running a "
              f"{n}-rank job on fewer devices (use --dry-launch to exercise the launch protocol on CPU)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL fails without it)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def dry_launch(args, world: int, rank: int) -> None:
    """The launch / timing protocol of the multi-GPU bench without devices: gloo process group, W warm-up + K timed
    'steps' (each = the 8-byte SUM all-reduce of the real step on a host scalar), barrier on both sides of the timed region,
    MAX over ranks, ONE JSON line from rank 0.  Covered by tests/test_host.py (world size 2, runs anywhere)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_data, m_ind, d_in, global_rows, strong, _ = WORKLOADS[args.workload]
    b_rows = global_rows // world if strong else global_rows
    acc = torch.zeros(1, dtype=torch.float64)

    def step(s):
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        acc[0] = t[0]
    for s in range(args.warmup):
        step(s)
    dist.barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(args.warmup + s)
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "svgp_elbo_steps_per_s", "value": None, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": float(el[0]) / max(args.steps, 1) * 1e3, "higher_is_better": True,
                          "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "none",
                          "dry_launch": True, "allreduce_check": float(acc[0]) == world * (world + 1) / 2.0,
                          "config": {"workload": f"launch protocol only ({args.workload}: M={m_ind}, D={d_in}, {b_rows} rows per rank)",
                                     "name": args.workload, "parallelism": f"dp{world}"}}))
    dist.barrier()
    dist.destroy_process_group()


def main():  # noqa: C901
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gpr", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and Python-mirror legs")
    ap.add_argument("--no-other", action="store_true", help="skip the other BASELINE configs (C3, C4 shards, C5)")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="N=1 only: bring up a world-size-1 nccl (RCCL) process group and put the 8-byte all-reduce of the "
                         "multi-GPU path behind every step -- what RCCL's own streams cost beside the library's")
    ap.add_argument("--dry-launch", action="store_true",
                    help="exercise ONLY the multi-rank launch protocol on CPU (gloo): rank spawn, rendezvous, barrier-bracketed "
                         "timing, MAX over ranks, the single JSON line from rank 0 -- no device work, `value` is null")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not under a launcher: become one.  `python bench.py --gpus N` starts N ranks itself (one process per GPU,
        # torch.distributed.run on 127.0.0.1, a free port) and rank 0 prints the single JSON line with n_gpus = N.
        raise SystemExit(spawn_ranks(args.gpus, args.dry_launch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it spawns the ranks itself)")
    if args.dry_launch:
        return dry_launch(args, world, rank)
    if torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} HIP devices, this node has {torch.cuda.device_count()}: refusing to "
                         f"run a {world}-rank job on fewer devices")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    from gpflow_amd import _lib, ops
    lib = _lib.load()
    selftest = args.rccl_selftest and world == 1
    # The library's internal streams are created by its first factorisation with n > 128, in an order chosen so that
    # chain and bulk streams land on different microengine pipes.  Let the library place its
    # streams FIRST: before RCCL brings up its own (multi-GPU runs) and before the host-to-device copies of the synthetic
    # inputs make the runtime open its copy queues.
    warm = torch.eye(256, dtype=torch.float64, device=device)
    ops.potrf_(warm, 256)
    torch.cuda.synchronize()
    del warm
    if world > 1 or selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    n_data, m_ind, d_in, global_rows, strong, seed = WORKLOADS[args.workload]
    if strong and global_rows % world:
        raise SystemExit("c4-strong needs a GPU count that divides 8192")
    b_rows = global_rows // world if strong else global_rows      # rows per rank per step
    rows_per_step = b_rows * world                                # rows of one global minibatch
    X, Y, Z, q_mu, q_sqrt, ls = make_inputs(n_data, m_ind, d_in, seed, device)
    ws = ops.svgp_elbo_workspace(m_ind, b_rows, d_in, P_LAT, False)
    out = torch.empty(2, dtype=torch.float64, device=device)
    info = torch.zeros(1, dtype=torch.int32, device=device)
    n_batches = n_data // rows_per_step
    scale = float(n_data) / float(rows_per_step)
    last = {}

    # the step's scalars (data term, KL) and the factorisation status land in pinned, device-mapped host memory by a
    # kernel store behind the step (gpk_publish_host); the host spins on the sequence word written last -- no blit copies,
    # no stream synchronise (round 3 timeline: ~130 us between the last kernel and the D2H blit, per step)
    mailbox = ops.HostMailbox(2)

    def shard_lo(s: int, r: int) -> int:
        return ((s % n_batches) * world + r) * b_rows  # rank r's shard of global minibatch s

    def step(s: int) -> float:
        lo = shard_lo(s, rank)
        ops.svgp_elbo_shard(Z, X[lo:lo + b_rows], Y[lo:lo + b_rows], q_mu, q_sqrt, variance=1.0, lengthscales=ls,
                            noise_variance=0.1, jitter=1e-6, ws=ws, out=out, info=info)
        if world > 1 or selftest:
            # RCCL over xGMI: one 8-byte all-reduce per step, enqueued behind the shard (no host sync in between)
            dist.all_reduce(out[0:1], op=dist.ReduceOp.SUM)
        mailbox.post(out, info)
        vals, inf = mailbox.wait()  # scalar is in host memory
        elbo = float(vals[0]) * scale - float(vals[1])
        last.update(elbo=elbo, info=inf, step=s)
        return elbo

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s)
    fence()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(args.warmup + s)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.cpu()[0])
    assert last["info"] == 0 and np.isfinite(last["elbo"]), last
    timed_last = dict(last)

    # ---- roofline leg (dominant kernel = the fp64 MFMA GEMM): HIP events around every GEMM launch ----
    roof = None
    nprof = 5
    if rank == 0:
        lib.gpk_profile_gemm_enable(1)
    for s in range(nprof):  # every rank steps (the all-reduce is collective); only rank 0 records
        step(args.warmup + args.steps + s)
    fence()
    if rank == 0:
        def by_kind(kind, min_flops=0.0):
            ms, n_launch, fl = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
            lib.gpk_profile_gemm_collect_kind(int(kind), ctypes.c_double(min_flops), ctypes.byref(ms), ctypes.byref(n_launch),
                                              ctypes.byref(fl))
            return ms.value, n_launch.value, fl.value

        def collect(min_flops, keep):
            ms, n_launch, fl = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
            lib.gpk_profile_gemm_collect_min(ctypes.c_double(min_flops), int(keep), ctypes.byref(ms),
                                             ctypes.byref(n_launch), ctypes.byref(fl))
            return ms.value, n_launch.value, fl.value
        kinds = {1: "gemm_nt_small", 2: "gemm_nt_fast<0,false>", 3: "gemm_nt_fast<0,true>", 4: "gemm_nt_fast<1,false>",
                 5: "gemm_nt_fast<1,true>", 6: "gemm_nt_kernel", 7: "svgp_step_kernel"}
        per_kernel = {}
        for kd, nm in kinds.items():
            ms_k, n_k, fl_k = by_kind(kd)
            if n_k:
                per_kernel[nm] = {"launches_per_step": n_k / nprof, "avg_launch_us": ms_k * 1e3 / n_k,
                                  "algorithmic_gflop_per_step": fl_k / nprof / 1e9,
                                  "tflops_over_summed_durations": fl_k / (ms_k * 1e-3) / 1e12 if ms_k > 0 else 0.0}
        # dominant kernel of a step = the kernel (template instantiation) carrying the most algorithmic flops; its
        # HIP-event average over ALL its launches is directly comparable with rocprofv3's per-kernel average (profiles/)
        dom = max(per_kernel, key=lambda k: per_kernel[k]["algorithmic_gflop_per_step"])
        dom_kind = [k for k, v in kinds.items() if v == dom][0]
        ms_dom, n_dom, fl_dom = by_kind(dom_kind)
        ms_big, n_big, fl_big = collect(1e9, True)
        ms_all, n_all, fl_all = collect(0.0, False)
        lib.gpk_profile_gemm_enable(0)
        ach = fl_dom / (ms_dom * 1e-3) / 1e12 if ms_dom > 0 else 0.0
        sink = torch.zeros(8, dtype=torch.float64, device=device)
        st = torch.cuda.current_stream().cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib.gpk_bench_mfma_f64(st, 512, 2000, sink.data_ptr()); torch.cuda.synchronize()
        e0.record(); lib.gpk_bench_mfma_f64(st, 512, 20000, sink.data_ptr()); e1.record(); torch.cuda.synchronize()
        ubench = 512 * 8 * 20000 * 8 * 2048 / (e0.elapsed_time(e1) * 1e-3) / 1e12
        traffic = None
        try:  # HBM bytes per launch of this kernel from the committed PMC passes (tools/profile_round.sh)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = float(json.load(f)[dom]["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
        step_tf = svgp_step_flops(m_ind, b_rows, P_LAT) * (args.steps / elapsed) / 1e12
        dom_desc = ("svgp_step_kernel: ONE persistent launch per step (factorisation chain + minibatch rows, 32 x 32 x 128 "
                    "v_mfma_f64_16x16x4_f64 slabs)") if dom == "svgp_step_kernel" else f"{dom}: v_mfma_f64_16x16x4_f64, 128x128x16 tiles"
        roof = {"bound": "mfma", "kernel": dom_desc,
                "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                "traffic": traffic, "launches_per_step": n_dom / nprof, "avg_launch_us": ms_dom * 1e3 / max(n_dom, 1),
                "algorithmic_gflop_per_launch": fl_dom / max(n_dom, 1) / 1e9,
                "step_level": {"achieved": step_tf, "frac": step_tf / FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "note": "algorithmic flops of the whole step (M^3/3 + 2 M^2 B) / wall-clock of the step "
                                       "(latency chain, launch gaps, reductions, D2H included)"},
                "per_kernel": per_kernel,
                "big_gemm_launches": {"launches_per_step": n_big / nprof, "avg_launch_us": ms_big * 1e3 / max(n_big, 1),
                                      "algorithmic_gflop_per_step": fl_big / nprof / 1e9,
                                      "tflops_over_summed_durations": fl_big / (ms_big * 1e-3) / 1e12 if ms_big > 0 else 0.0},
                "all_gemm_launches": {"launches_per_step": n_all / nprof, "avg_launch_us": ms_all * 1e3 / max(n_all, 1),
                                      "algorithmic_gflop_per_step": fl_all / nprof / 1e9,
                                      "tflops_over_summed_durations": fl_all / (ms_all * 1e-3) / 1e12 if ms_all > 0 else 0.0},
                "mfma_f64_issue_ubench_tflops": ubench,
                "frac_of_measured_mfma_ceiling": ach / ubench if ubench > 0 else None,
                "note": "achieved = algorithmic flops (triangular K ranges counted once) of ALL launches of the dominant "
                        "kernel in a step / the sum of their HIP-event durations, events recorded on each launch's own "
                        "stream; peak = AMD datasheet FP64 matrix (the in-image guide lists no fp64 peak; the "
                        "v_mfma_f64_16x16x4 issue rate measured on this chip is next to it); traffic = FETCH_SIZE x2 "
                        "(gfx950 correction for 16-B coalesced loads) + WRITE_SIZE of the same kernel from the PMC passes "
                        "in profiles/ (MALL hits included, so an upper bound on HBM bytes)"}

    if world > 1:
        dist.barrier()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    steps_per_s = args.steps / elapsed
    value = steps_per_s if strong else steps_per_s * world
    names = {"cm": "BASELINE metric config Cm", "c3": "BASELINE config C3", "c4-weak": "BASELINE config C4, weak scaling",
             "c4-strong": "BASELINE config C4, strong scaling"}
    res = {
        "metric": "svgp_elbo_steps_per_s", "value": value, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"SVGP RBF(ARD)+Gaussian ELBO, N={n_data:.0e} M={m_ind} D={d_in} P=1 whitened, "
                               f"{b_rows} minibatch rows per GPU per step ({names[args.workload]})",
                   "name": args.workload, "rows_per_gpu_per_step": b_rows, "global_batch": rows_per_step,
                   "parallelism": f"dp{world} (minibatch rows sharded, Z/q replicated, one 8-byte RCCL all-reduce per step)",
                   "value_counts": "global steps/s" if strong else "8192-row minibatch evaluations/s over the whole job"},
        "global_steps_per_s": steps_per_s, "last_elbo": timed_last["elbo"],
        "step_tflops_per_gpu": svgp_step_flops(m_ind, b_rows, P_LAT) * steps_per_s / 1e12,
        "step_frac_of_fp64_peak": svgp_step_flops(m_ind, b_rows, P_LAT) * steps_per_s / 1e12 / FP64_PEAK_TFLOPS,
        "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"),
        "library": lib.gpk_version().decode(),
        "stream_selfcheck": stream_selfcheck(lib),
        "roofline": roof,
    }
    if selftest:
        res["rccl_selftest"] = ("world-size-1 nccl process group: every step above carries the 8-byte all-reduce (RCCL "
                                "initialised AFTER the library placed its streams); compare ms_per_step with a run without the flag")
    if not args.no_cpu_baseline:
        # parity of the LAST TIMED step against the oracle on the same arrays (all shards of that global minibatch; the
        # data are replicated on every rank), and the oracle's own wall-clock as the CPU baseline
        s_last = timed_last["step"]
        idx = torch.cat([torch.arange(shard_lo(s_last, r), shard_lo(s_last, r) + b_rows, device=device) for r in range(world)])
        base, ref, err = cpu_baseline_and_parity(X[idx].cpu().numpy(), Y[idx].cpu().numpy(), Z.cpu().numpy(),
                                                 q_mu.cpu().numpy(), q_sqrt.cpu().numpy(), ls, n_data, timed_last["elbo"])
        if strong or world > 1:
            base["sample"] += f" (global minibatch of {rows_per_step} rows)"
        res["cpu_baseline"] = base
        res["oracle_elbo"] = ref
        res["parity_rel_err"] = err
        res["parity_ok"] = bool(err <= 1e-8)
    if world == 1 and not args.no_extras:
        # PCIe-inclusive rate (never `value`): the same step when the minibatch arrives in (pinned) HOST memory, as it
        # does for a caller handing NumPy arrays to the Python mirror -- rows x (D + 1) doubles per step
        hX = [X[i * b_rows:(i + 1) * b_rows].cpu().pin_memory() for i in range(4)]
        hY = [Y[i * b_rows:(i + 1) * b_rows].cpu().pin_memory() for i in range(4)]
        dX, dY = torch.empty_like(X[:b_rows]), torch.empty_like(Y[:b_rows])

        def host_step(s):
            dX.copy_(hX[s % 4], non_blocking=True)
            dY.copy_(hY[s % 4], non_blocking=True)
            ops.svgp_elbo_shard(Z, dX, dY, q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6,
                                ws=ws, out=out, info=info)
            mailbox.post(out, info)
            mailbox.wait()
        for s in range(3):
            host_step(s)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for s in range(30):
            host_step(s)
        torch.cuda.synchronize()
        res["pcie_inclusive_steps_per_s"] = 30.0 / (time.perf_counter() - t1)
        # the same steps with the read-back one step late (never `value`): step s + 1 is enqueued before the host waits for
        # the scalars of step s -- what a monitoring loop that does not feed the value back can do; it removes the host's
        # turn-around (wait, then ~80 enqueues before the first kernel of the next step) from the GPU's critical path
        boxes = [ops.HostMailbox(2), ops.HostMailbox(2)]

        def piped(n):
            for s in range(n):
                lo = shard_lo(s, 0)
                ops.svgp_elbo_shard(Z, X[lo:lo + b_rows], Y[lo:lo + b_rows], q_mu, q_sqrt, variance=1.0, lengthscales=ls,
                                    noise_variance=0.1, jitter=1e-6, ws=ws, out=out, info=info)
                boxes[s & 1].post(out, info)
                if s:
                    boxes[(s - 1) & 1].wait()
            return boxes[(n - 1) & 1].wait()
        piped(4)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pv, pinf = piped(40)
        torch.cuda.synchronize()
        res["readback_one_step_late_steps_per_s"] = 40.0 / (time.perf_counter() - t1)
        assert pinf == 0 and np.isfinite(pv[0])
        # the same step through the Python mirror (gpflow_amd.models.SVGP.elbo: Parameters with cached device values,
        # hyper-parameters converted on the host every call, ctypes into the same fused driver, float() of the result)
        import gpflow_amd as gpflow
        model = gpflow.models.SVGP(gpflow.kernels.SquaredExponential(variance=1.0, lengthscales=ls),
                                   gpflow.likelihoods.Gaussian(0.1), Z.cpu().numpy(), q_mu=q_mu.cpu().numpy(),
                                   q_sqrt=q_sqrt.cpu().numpy(), num_data=n_data)
        for s in range(3):
            v = float(model.elbo((X[s * b_rows:(s + 1) * b_rows], Y[s * b_rows:(s + 1) * b_rows])))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for s in range(30):
            lo = (s % n_batches) * b_rows
            v = float(model.elbo((X[lo:lo + b_rows], Y[lo:lo + b_rows])))
        res["python_mirror_steps_per_s"] = 30.0 / (time.perf_counter() - t1)
        res["python_mirror_last_elbo"] = v
    if world == 1 and not args.no_train:
        res["train_step"] = train_step_leg(X, Y, Z, q_mu, q_sqrt, ls, n_data, b_rows)
    if world == 1 and not args.no_other:
        X = Y = None
        torch.cuda.empty_cache()
        res["other_workloads"] = other_workloads_leg(device, with_oracle=not args.no_cpu_baseline)
    if world == 1 and not args.no_gpr:
        X = Y = None
        torch.cuda.empty_cache()
        res["gpr_cholesky"] = gpr_leg(ops, lib, device, with_oracle=not args.no_cpu_baseline)
    print(json.dumps(res))
    if world > 1 or selftest:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
