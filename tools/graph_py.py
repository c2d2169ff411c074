import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import _lib, ops
lib = _lib.load()
mode = sys.argv[1]
n, extra = 2048, 8192
K = 0.01 * np.ones((n + extra, n)); K[np.arange(n), np.arange(n)] = 10.0
dev = ops.device()
if mode == "torchstream":
    st = torch.cuda.Stream(device=dev)
elif mode == "rawstream":
    hip = ctypes.CDLL("libamdhip64.so")
    raw = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(raw)) == 0
    st = torch.cuda.ExternalStream(raw.value, device=dev)
else:
    st = torch.cuda.current_stream()
with torch.cuda.stream(st):
    T = ops.to_device(K)
    invd = ops.invd_alloc(n)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(4):
        print("call", it, flush=True)
        rc = lib.gpk_potrf(torch.cuda.current_stream().cuda_stream, T.data_ptr(), n, extra, n, 1, 0, invd.data_ptr(), 0, info.data_ptr())
        torch.cuda.synchronize()
        print("  rc", rc, float(T[0, 0]), flush=True)
print("ok", mode)
