"""bench.py -- headline benchmark of the dense-GP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): SVGP ELBO steps/s at N=1e6, M=2048, D=8 (config "Cm"), with the GPR Cholesky at
N=16384 (GF/s vs fp64 peak) reported alongside in the same JSON line (key "gpr_cholesky", N=1 only).

A "step" = one forward minibatch ELBO evaluation (SVGP.elbo, gpflow/models/svgp.py:166-181) over
B = 8192 rows per GPU: Kuu / Kuf builds, Cholesky of Kuu, the triangular solves, the q_sqrt projection,
the variational expectations, KL, the (multi-GPU) all-reduce of the per-shard data term and the scalar
landing in host memory.  Weak scaling: every rank keeps the same 8192-row shard size, so a global step
covers 8192*N rows and `value` = N * (global steps / s) = 8192-row minibatch evaluations per second
over the whole job.  Inputs (the 1e6 x 8 data matrix, Z, q) are resident in HBM before the timed region.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The factorisation runs a latency-critical
# panel stream next to bulk streams; with 2 hardware queues the SVGP step measured 415 steps/s vs 380 with 4 and
# 350 with 8 (A/B on one MI355X, tools/ab.sh).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_DATA, M_IND, D_IN, B_ROWS, P_LAT = 1_000_000, 2048, 8, 8192, 1
FP64_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet, FP64 matrix (= FP64 vector); the in-image guide lists no fp64 row


def svgp_step_flops(m: int, b: int, p: int) -> float:
    """Algorithmic flops of one whitened step: M^3/3 + M^2 B (1 + P)."""
    return m ** 3 / 3.0 + float(m) * m * b * (1 + p)


def make_inputs(rank: int, device):
    """SURVEY 8d config Cm: X ~ N(0,1) seed 4, Y = sin(sum x) + 0.1 eps, Z = first M rows + 0.01 noise,
    q_mu ~ 0.1 N(0,1), q_sqrt = tril(0.05 N(0,1)) + 0.5 I, ARD lengthscales sqrt(D)(0.8 + 0.05 d), noise 0.1.
    The data are pre-shuffled once (seed 5) so minibatch s of rank r is a contiguous slice."""
    g = torch.Generator(device="cpu").manual_seed(4)
    X = torch.randn((N_DATA, D_IN), generator=g, dtype=torch.float64)
    Y = torch.sin(X.sum(1, keepdim=True)) + 0.1 * torch.randn((N_DATA, P_LAT), generator=g, dtype=torch.float64)
    Z = X[:M_IND] + 0.01 * torch.randn((M_IND, D_IN), generator=g, dtype=torch.float64)
    q_mu = 0.1 * torch.randn((M_IND, P_LAT), generator=g, dtype=torch.float64)
    q_sqrt = torch.tril(0.05 * torch.randn((P_LAT, M_IND, M_IND), generator=g, dtype=torch.float64)) \
        + 0.5 * torch.eye(M_IND, dtype=torch.float64)
    perm = torch.randperm(N_DATA, generator=torch.Generator(device="cpu").manual_seed(5))
    X, Y = X[perm], Y[perm]
    ls = np.sqrt(D_IN) * (0.8 + 0.05 * np.arange(D_IN))
    return (X.to(device), Y.to(device), Z.to(device).contiguous(), q_mu.to(device).contiguous(),
            q_sqrt.to(device).contiguous(), ls)


def cpu_baseline(budget_s: float = 12.0):
    """The oracle (NumPy/SciPy restatement of GPflow's algorithm; TensorFlow itself is not installable
    here) timed on this box's host cores on the SAME step: M=2048, B=8192, D=8, whitened, P=1."""
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(4)
    X = rng.normal(size=(B_ROWS * 2, D_IN))
    Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.normal(size=(B_ROWS * 2, 1))
    Z = X[:M_IND] + 0.01 * rng.normal(size=(M_IND, D_IN))
    q_mu = 0.1 * rng.normal(size=(M_IND, 1))
    q_sqrt = (np.tril(0.05 * rng.normal(size=(M_IND, M_IND))) + 0.5 * np.eye(M_IND))[None]
    ls = np.sqrt(D_IN) * (0.8 + 0.05 * np.arange(D_IN))
    kw = dict(variance=1.0, lengthscales=ls, noise_variance=0.1, whiten=True, num_data=N_DATA)
    orc.svgp_elbo(X[:B_ROWS], Y[:B_ROWS], Z, q_mu, q_sqrt, **kw)  # warm-up (benchmark/run.py:71 convention)
    times, t_start = [], time.perf_counter()
    while time.perf_counter() - t_start < budget_s and len(times) < 50:
        s = len(times) % 2
        t0 = time.perf_counter()
        orc.svgp_elbo(X[s * B_ROWS:(s + 1) * B_ROWS], Y[s * B_ROWS:(s + 1) * B_ROWS], Z, q_mu, q_sqrt, **kw)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    try:
        import threadpoolctl
        threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": 1.0 / med, "unit": "steps/s", "cores": int(threads), "kind": "port",
            "sample": f"{len(times)} ELBO steps of the workload (M={M_IND}, B={B_ROWS}, D={D_IN}, P=1, whitened), "
                      f"median {med * 1e3:.1f} ms/step, NumPy/SciPy (OpenBLAS) oracle; GPflow+TensorFlow is not "
                      f"installable in this image"}


def train_step_leg(X, Y, Z, q_mu, q_sqrt, ls, steps: int = 20):
    """SURVEY 8f row 1 (the caller of the hot path): one TRAINING step = forward + hand-written reverse pass
    (gpflow_amd/gradients.py) + Adam update, same config Cm, reported beside the headline ELBO metric (not part of it)."""
    from gpflow_amd import gradients
    kw = dict(variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, scale=float(N_DATA) / B_ROWS)
    n_batches = N_DATA // B_ROWS
    m = {k: torch.zeros_like(v) for k, v in (("Z", Z), ("q_mu", q_mu), ("q_sqrt", q_sqrt))}
    v2 = {k: torch.zeros_like(v) for k, v in m.items()}
    par = {"Z": Z.clone(), "q_mu": q_mu.clone(), "q_sqrt": q_sqrt.clone()}

    def one(s):
        lo = (s % n_batches) * B_ROWS
        F, g, info = gradients.svgp_elbo_and_grad(par["Z"], X[lo:lo + B_ROWS], Y[lo:lo + B_ROWS], par["q_mu"],
                                                  par["q_sqrt"], **kw)
        for k in par:  # Adam (tf.keras defaults) on the device-resident variables
            m[k].mul_(0.9).add_(g[k], alpha=-0.1)
            v2[k].mul_(0.999).addcmul_(g[k], g[k], value=0.001)
            par[k].addcdiv_(m[k], v2[k].sqrt().add_(1e-7), value=-1e-3)
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
    flops = 3.0 * svgp_step_flops(M_IND, B_ROWS, P_LAT)  # forward + ~2x for the reverse pass (same GEMM shapes)
    return {"workload": "SVGP training step (ELBO + gradients w.r.t. Z, q_mu, q_sqrt, kernel and noise parameters + Adam), "
                        "config Cm, 8192 rows", "ms_per_step": dt * 1e3, "steps_per_s": 1.0 / dt, "last_elbo": elbo,
            "info": info, "approx_tflops": flops / dt / 1e12,
            "note": "hyper-parameters held fixed in this leg (their gradients are computed and read back); "
                    "SVGPTrainer updates them on the host"}


def gpr_cholesky_leg(ops, lib, device):  # noqa: C901
    """GPR config C2: K(X,X)+noise build + Cholesky + LML tail at N=16384, D=8 (gpr.py:91-107)."""
    n, d = 16384, 8
    g = torch.Generator(device="cpu").manual_seed(2)
    X = torch.randn((n, d), generator=g, dtype=torch.float64).to(device)
    Y = (torch.sin(X.sum(1, keepdim=True)) + 0.1 * torch.randn((n, 1), dtype=torch.float64, device=device))
    ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
    ws = torch.empty(int(lib.gpk_gpr_lml_workspace_bytes(n, d, 1)) // 8 + 1, dtype=torch.float64, device=device)
    kw = dict(variance=1.0, lengthscales=ls, noise_variance=0.1, ws=ws)
    for _ in range(2):
        out, info = ops.gpr_lml(X, Y, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out, info = ops.gpr_lml(X, Y, **kw); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.median(ts))
    K = torch.empty((n, n), dtype=torch.float64, device=device)
    tk = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.kernel_matrix(X, None, variance=1.0, lengthscales=ls, diag_add=0.1, out=K); e1.record()
        torch.cuda.synchronize(); tk.append(e0.elapsed_time(e1) * 1e-3)
    tkb = float(np.min(tk[1:]))
    flops = n ** 3 / 3.0
    kb_alg = n * n * 8 + n * d * 8  # algorithmic bytes: the full N x N fp64 write + the N x D read
    # the trailing update on its own (north_star: ">= 60 % of fp64 MFMA peak on the N=16384 Cholesky trailing update"):
    # HIP events around every GEMM launch of one more factorisation; the outer rest-updates  A22 -= P P^T  (lower tiles,
    # K = 768) are the launches with >= 2e10 algorithmic flop (look-ahead strips and panel-internal GEMMs are smaller)
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
    trailing = {"bound": "mfma", "kernel": "gemm_nt_fast<0,false>, lower tiles, K = 768 (outer trailing updates)",
                "achieved": tu_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tu_tf / FP64_PEAK_TFLOPS,
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
                        "stream (CU-masked: 240 of 256 CUs; the look-ahead panel runs beside them on the other 16)"}
    return {"workload": "GPR RBF N=16384 D=8 fp64: K build + Cholesky + LML (one gpk_gpr_lml call)",
            "kernel_build_roofline": {"bound": "hbm", "kernel": "rbf_kernel (full N x N)", "achieved": kb_alg / tkb / 1e9,
                                      "peak": 8000.0, "unit": "GB/s", "frac": kb_alg / tkb / 8e12, "traffic": None},
            "trailing_update_roofline": trailing,
            "lml": float(out.cpu()[0]), "info": int(info.cpu()[0]), "ms_total": t * 1e3,
            "cholesky_gflops_incl_build_and_tail": flops / t / 1e9,
            "frac_of_fp64_peak": flops / t / 1e12 / FP64_PEAK_TFLOPS,
            "kernel_build_full_ms": tkb * 1e3, "kernel_build_full_GBps": n * n * 8 / tkb / 1e9,
            "kernel_build_frac_of_8TBps": n * n * 8 / tkb / 8e12}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gpr", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from gpflow_amd import _lib, ops
    lib = _lib.load()
    X, Y, Z, q_mu, q_sqrt, ls = make_inputs(rank, device)
    ws = ops.svgp_elbo_workspace(M_IND, B_ROWS, D_IN, P_LAT, False)
    out = torch.empty(2, dtype=torch.float64, device=device)
    info = torch.zeros(1, dtype=torch.int32, device=device)
    n_batches = N_DATA // (B_ROWS * world)
    scale = float(N_DATA) / float(B_ROWS * world)
    last = {}

    # the step's scalars land in pinned host memory by two async copies + one stream synchronise
    h_out = torch.empty(2, dtype=torch.float64).pin_memory()
    h_info = torch.empty(1, dtype=torch.int32).pin_memory()

    def step(s: int) -> float:
        lo = ((s % n_batches) * world + rank) * B_ROWS  # this rank's shard of global minibatch s
        ops.svgp_elbo_shard(Z, X[lo:lo + B_ROWS], Y[lo:lo + B_ROWS], q_mu, q_sqrt, variance=1.0, lengthscales=ls,
                            noise_variance=0.1, jitter=1e-6, ws=ws, out=out, info=info)
        if world > 1:
            dist.all_reduce(out[0:1], op=dist.ReduceOp.SUM)  # RCCL over xGMI: one 8-byte all-reduce per step
        h_out.copy_(out, non_blocking=True)
        h_info.copy_(info, non_blocking=True)
        torch.cuda.current_stream().synchronize()  # scalar is in host memory
        elbo = float(h_out[0]) * scale - float(h_out[1])
        last.update(elbo=elbo, info=int(h_info[0]))
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

    # ---- roofline leg (dominant kernel = the fp64 MFMA GEMM): HIP events around every GEMM launch ----
    roof = None
    nprof = 5
    if rank == 0:
        lib.gpk_profile_gemm_enable(1)
    for s in range(nprof):  # every rank steps (the all-reduce is collective); only rank 0 records
        step(args.warmup + args.steps + s)
    fence()
    if rank == 0:
        def collect(min_flops, keep):
            ms, n_launch, fl = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
            lib.gpk_profile_gemm_collect_min(ctypes.c_double(min_flops), int(keep), ctypes.byref(ms),
                                             ctypes.byref(n_launch), ctypes.byref(fl))
            return ms.value, n_launch.value, fl.value
        # dominant kernel of a step = the q_sqrt projection launch of gemm_nt_fast (EPI=1, paired triangular-K
        # tiles): the only launch with >= 3e10 algorithmic flop, one per step, fixed shape -> its HIP-event
        # duration is directly comparable with rocprofv3's average for `gemm_nt_fast<1, true>` (profiles/).
        ms_dom, n_dom, fl_dom = collect(3e10, True)
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
                traffic = float(json.load(f)["gemm_nt_fast<1,true>"]["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
        roof = {"bound": "mfma", "kernel": "gemm_nt_fast<1,true>: q_sqrt projection, v_mfma_f64_16x16x4_f64, 128x128x16 tiles",
                "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                "traffic": traffic, "launches_per_step": n_dom / nprof, "avg_launch_us": ms_dom * 1e3 / max(n_dom, 1),
                "algorithmic_gflop_per_launch": fl_dom / max(n_dom, 1) / 1e9,
                "big_gemm_launches": {"launches_per_step": n_big / nprof, "avg_launch_us": ms_big * 1e3 / max(n_big, 1),
                                      "algorithmic_gflop_per_step": fl_big / nprof / 1e9,
                                      "tflops_over_summed_durations": fl_big / (ms_big * 1e-3) / 1e12 if ms_big > 0 else 0.0},
                "all_gemm_launches": {"launches_per_step": n_all / nprof, "avg_launch_us": ms_all * 1e3 / max(n_all, 1),
                                      "algorithmic_gflop_per_step": fl_all / nprof / 1e9,
                                      "tflops_over_summed_durations": fl_all / (ms_all * 1e-3) / 1e12 if ms_all > 0 else 0.0},
                "mfma_f64_issue_ubench_tflops": ubench,
                "frac_of_measured_mfma_ceiling": ach / ubench if ubench > 0 else None,
                "note": "achieved = algorithmic flops of the launch (2 * B * sum over column tiles of the non-zero K range: "
                        "M^2 B with the triangle of q_sqrt counted once) / its HIP-event duration, events recorded on the "
                        "launch stream; peak = AMD datasheet FP64 matrix (the in-image guide lists no fp64 peak; the "
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
    value = steps_per_s * world
    res = {
        "metric": "svgp_elbo_steps_per_s", "value": value, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "SVGP RBF(ARD)+Gaussian ELBO, N=1e6 M=2048 D=8 P=1 whitened, minibatch 8192 rows per GPU "
                               "(BASELINE metric config Cm)", "rows_per_gpu_per_step": B_ROWS,
                   "global_batch": B_ROWS * world, "parallelism": f"dp{world} (minibatch rows sharded, Z/q replicated, "
                                                                  f"one 8-byte RCCL all-reduce per step)"},
        "global_steps_per_s": steps_per_s, "last_elbo": last["elbo"],
        "step_tflops_per_gpu": svgp_step_flops(M_IND, B_ROWS, P_LAT) * steps_per_s / 1e12,
        "step_frac_of_fp64_peak": svgp_step_flops(M_IND, B_ROWS, P_LAT) * steps_per_s / 1e12 / FP64_PEAK_TFLOPS,
        "roofline": roof,
    }
    if world == 1:
        # PCIe-inclusive rate (never `value`): the same step when the minibatch arrives in (pinned) HOST memory, as it
        # does for a caller handing NumPy arrays to the Python mirror -- 8192 x (8 + 1) doubles = 0.59 MB per step
        hX = [X[i * B_ROWS:(i + 1) * B_ROWS].cpu().pin_memory() for i in range(4)]
        hY = [Y[i * B_ROWS:(i + 1) * B_ROWS].cpu().pin_memory() for i in range(4)]
        dX, dY = torch.empty_like(X[:B_ROWS]), torch.empty_like(Y[:B_ROWS])

        def host_step(s):
            dX.copy_(hX[s % 4], non_blocking=True)
            dY.copy_(hY[s % 4], non_blocking=True)
            ops.svgp_elbo_shard(Z, dX, dY, q_mu, q_sqrt, variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6,
                                ws=ws, out=out, info=info)
            h_out.copy_(out, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        for s in range(3):
            host_step(s)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for s in range(30):
            host_step(s)
        torch.cuda.synchronize()
        res["pcie_inclusive_steps_per_s"] = 30.0 / (time.perf_counter() - t1)
    if world == 1 and not args.no_train:
        res["train_step"] = train_step_leg(X, Y, Z, q_mu, q_sqrt, ls)
    if world == 1 and not args.no_gpr:
        res["gpr_cholesky"] = gpr_cholesky_leg(ops, lib, device)
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline()
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
