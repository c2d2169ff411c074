"""Time ops.combine_parts on the shapes of the training step (8 x 2048 x 2048 lower; 32 x 2048 x 17) against torch.sum."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops
dev = ops.device()
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for shape, lower in (((8, 2048, 2048), True), ((8, 2048, 2048), False), ((32, 2048, 17), False), ((2, 2048, 2048), True)):
    P = torch.randn(shape, dtype=torch.float64, device=dev)
    a = t(lambda: ops.combine_parts(P, lower=lower))
    b = t(lambda: (torch.tril(P.sum(0)) if lower else P.sum(0)))
    ref = torch.tril(P.sum(0)) if lower else P.sum(0)
    err = float((ops.combine_parts(P, lower=lower) - ref).abs().max())
    print("parts %-16s lower=%-5s combine_parts %7.1f us   torch %7.1f us   max |diff| %.1e" % (shape, lower, a, b, err))
