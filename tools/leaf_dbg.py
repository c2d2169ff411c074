"""Phase timing of the 128x128 leaf kernel (s_memrealtime ticks, 10 ns) + correctness of L and L^-1."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import _lib, ops
lib = _lib.load()
lib.gpk_debug_set_leaf_timing.argtypes = [ctypes.c_void_p]
dbg = torch.zeros(8, dtype=torch.int64, device="cuda")
lib.gpk_debug_set_leaf_timing(dbg.data_ptr())
rng = np.random.default_rng(0)
for n in (128, 100, 16):
    X = rng.normal(size=(n, 3))
    K = np.exp(-0.5 * ((X[:, None] - X[None]) ** 2).sum(-1)) + 0.1 * np.eye(n)
    for it in range(3):
        T = ops.to_device(K)
        invd, info = ops.potrf_(T, n)
        torch.cuda.synchronize()
        v = dbg.cpu().numpy()
    L = np.tril(T.cpu().numpy())
    Lr = np.linalg.cholesky(K)
    Xi = invd.cpu().numpy().reshape(128, 128)[:n, :n]
    print(f"n={n} info={int(info[0])} ticks(10ns): load {v[0]} factor {v[1]} inverse {v[2]} store {v[3]} total {v[4]}  "
          f"|L-Lref|={np.abs(L - Lr).max():.2e} |X L - I|={np.abs(Xi @ Lr - np.eye(n)).max():.2e}")
