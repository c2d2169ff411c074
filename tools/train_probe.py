"""SVGP training step (config Cm) under the microscope: ms per step for a few host-side variants, then a per-call table
(HIP events around every gpflow_amd.ops call of one reverse-mode evaluation, aggregated by op + shapes + flags) and the
time that is NOT inside any library call (torch glue + launch gaps).  Also the GPR value + gradient at N = GPR_N.

    python tools/train_probe.py [table] [gpr]
Variants are chosen with TRAIN_VARIANTS="name,name" (default: all); the library with GPK_LIBRARY as usual."""
import os, sys, time, collections
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from gpflow_amd import ops, gradients

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n_data, m_ind, d_in, b_rows, _, seed = bench.WORKLOADS["cm"]
X, Y, Z, q_mu, q_sqrt, ls = bench.make_inputs(n_data, m_ind, d_in, seed, dev)
kw = dict(variance=1.0, lengthscales=ls, noise_variance=0.1, jitter=1e-6, scale=float(n_data) / b_rows)
nb = n_data // b_rows


def run(steps, warm=3):
    out = None
    for s in range(warm):
        out = gradients.svgp_elbo_and_grad(Z, X[:b_rows], Y[:b_rows], q_mu, q_sqrt, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        lo = (s % nb) * b_rows
        out = gradients.svgp_elbo_and_grad(Z, X[lo:lo + b_rows], Y[lo:lo + b_rows], q_mu, q_sqrt, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


variants = {
    "default": {},
    "no_overlap": {"OVERLAP_BRANCHES": False},
}
want = [v for v in os.environ.get("TRAIN_VARIANTS", ",".join(variants)).split(",") if v]
ref = None
for name in want:
    saved = {k: getattr(gradients, k) for k in variants[name]}
    for k, v in variants[name].items():
        setattr(gradients, k, v)
    ms, (F, g, info) = run(20)
    ms2, _ = run(20, warm=0)
    chk = float(F.cpu()[0]), float(g["q_sqrt"].abs().sum().cpu()), float(g["Z"].abs().sum().cpu()), float(g["lengthscales"].abs().sum().cpu())
    if ref is None:
        ref = chk
    rel = max(abs(a - b) / max(abs(b), 1e-300) for a, b in zip(chk, ref))
    print(f"train variant={name:12s} lib={os.path.basename(os.environ.get('GPK_LIBRARY', 'libgpk.so'))} "
          f"tune=[{' '.join(k + '=' + v for k, v in os.environ.items() if k.startswith('GPK_') and k != 'GPK_LIBRARY')}] "
          f"ms/step {ms:.3f} {ms2:.3f}  F={chk[0]:.6f} rel_vs_first={rel:.2e} info={int(info.cpu()[0])}", flush=True)
    for k, v in saved.items():
        setattr(gradients, k, v)

if "table" in sys.argv:
    gradients.OVERLAP_BRANCHES = False   # one stream: the events then bracket exactly one call each
    recs = []

    def wrap(name, fn):
        def f(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            shp = [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)]
            flags = {kk: vv for kk, vv in k.items() if kk in ("b_tri", "c_lower", "mode", "identity_rows", "beta")}
            recs.append((name, str(shp), str(flags), e0, e1))
            return r
        return f

    names = ["kernel_matrix", "kernel_matrix_hadamard", "potrf_", "gemm_nt", "transpose", "row_stats", "gaussian_varexp_sum",
             "gauss_kl_white"]
    orig = {n: getattr(ops, n) for n in names}
    for n in names:
        setattr(ops, n, wrap(n, orig[n]))
    tot = []
    for rep in range(4):
        del recs[:]
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        gradients.svgp_elbo_and_grad(Z, X[:b_rows], Y[:b_rows], q_mu, q_sqrt, **kw)
        a1.record()
        torch.cuda.synchronize()
        tot.append(a0.elapsed_time(a1))
    for n in names:
        setattr(ops, n, orig[n])
    agg = collections.OrderedDict()
    for name, shp, flags, e0, e1 in recs:
        key = (name, shp, flags)
        agg.setdefault(key, [0, 0.0])
        agg[key][0] += 1
        agg[key][1] += e0.elapsed_time(e1)
    inside = sum(v[1] for v in agg.values())
    print(f"one evaluation (single stream, events on): {tot[-1]:.3f} ms; inside library calls {inside:.3f} ms; "
          f"outside (torch glue + gaps) {tot[-1] - inside:.3f} ms")
    for (name, shp, flags), (cnt, ms) in agg.items():
        print(f"  {ms * 1e3 / cnt:9.1f} us x{cnt:2d}  {name:24s} {shp} {flags}")

if "gpr" in sys.argv:
    n = int(os.environ.get("GPR_N", "16384"))
    rng = np.random.default_rng(3)
    Xg = ops.to_device(rng.normal(size=(n, 8)))
    Yg = ops.to_device(rng.normal(size=(n, 1)))
    lsg = np.sqrt(8) * np.ones(8)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lml, gg, info = gradients.gpr_lml_and_grad(Xg, Yg, variance=1.0, lengthscales=lsg, noise_variance=0.1)
        torch.cuda.synchronize()
        print(f"gpr value+grad N={n}: {(time.perf_counter() - t0) * 1e3:.1f} ms  lml={float(lml.cpu()[0]):.6f} "
              f"dvar={float(gg['variance'].cpu()[0]):.6e} info={int(info.cpu()[0])}", flush=True)
