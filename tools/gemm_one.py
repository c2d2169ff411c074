"""One GEMM shape a few times (for rocprofv3 --pmc passes): python tools/gemm_one.py m n k beta lower reps"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops
m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
beta, low, reps = float(sys.argv[4]), bool(int(sys.argv[5])), int(sys.argv[6])
dev = ops.device()
g = torch.Generator(device="cpu").manual_seed(0)
A = torch.randn((m, k), generator=g, dtype=torch.float64).to(dev)
B = A if m == n else torch.randn((n, k), generator=g, dtype=torch.float64).to(dev)
C = torch.zeros((m, n), dtype=torch.float64, device=dev)
for _ in range(reps):
    ops.gemm_nt(A, B, alpha=-1.0, beta=beta, C=C, c_lower=low)
torch.cuda.synchronize()
