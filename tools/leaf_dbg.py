import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import _lib, ops
lib = _lib.load()
lib.gpk_debug_set_leaf_timing.argtypes = [ctypes.c_void_p]
dbg = torch.zeros(8, dtype=torch.int64, device="cuda")
lib.gpk_debug_set_leaf_timing(dbg.data_ptr())
rng = np.random.default_rng(0)
n = 128
X = rng.normal(size=(n, 3))
K = np.exp(-0.5 * ((X[:, None] - X[None]) ** 2).sum(-1)) + 0.1 * np.eye(n)
sink = torch.zeros(8, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for it in range(4):
    if it >= 2:
        lib.gpk_bench_mfma_f64(st, 1024, 40000, sink.data_ptr())  # ~50 ms of fp64 MFMA right before
    T = ops.to_device(K)
    ops.potrf_(T, n)
    torch.cuda.synchronize()
    v = dbg.cpu().numpy()
    print("ticks(10ns): load", v[0], "A", v[1], "B", v[2], "C", v[3], "loop", v[4], "total_after_store", v[5], "clock64 ticks", v[6], "ratio GHz", v[6] / (v[5] * 10e-9) / 1e9)
