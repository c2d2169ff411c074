"""Per-call table of GPR value + gradient at N = GPR_N (default 16384): HIP events around every gpflow_amd.ops call of
gradients.gpr_lml_and_grad; the remainder is torch glue."""
import os, sys, time, collections
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops, gradients
n = int(os.environ.get("GPR_N", "16384"))
rng = np.random.default_rng(3)
Xg = ops.to_device(rng.normal(size=(n, 8))); Yg = ops.to_device(rng.normal(size=(n, 1)))
lsg = np.sqrt(8) * np.ones(8)
kw = dict(variance=1.0, lengthscales=lsg, noise_variance=0.1)
gradients.gpr_lml_and_grad(Xg, Yg, **kw); torch.cuda.synchronize()
recs = []
def wrap(name, fn):
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        recs.append((name, str([tuple(x.shape) for x in a if isinstance(x, torch.Tensor)]), str({kk: vv for kk, vv in k.items() if kk in ("b_tri", "c_lower", "mode", "identity_rows", "beta", "lower")}), e0, e1))
        return r
    return f
names = ["kernel_matrix", "kernel_matrix_hadamard", "potrf_", "gemm_nt", "transpose", "row_stats", "combine_parts"]
for nm in names: setattr(ops, nm, wrap(nm, getattr(ops, nm)))
a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a0.record(); lml, g, info = gradients.gpr_lml_and_grad(Xg, Yg, **kw); a1.record(); torch.cuda.synchronize()
tot = a0.elapsed_time(a1); inside = sum(e0.elapsed_time(e1) for *_, e0, e1 in recs)
print("GPR value+grad N=%d: %.1f ms; inside library calls %.1f ms; torch glue + gaps %.1f ms" % (n, tot, inside, tot - inside))
for name, shp, fl, e0, e1 in recs:
    print("  %9.2f ms  %-24s %s %s" % (e0.elapsed_time(e1), name, shp, fl))
