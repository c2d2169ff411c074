"""Which part of the library survives hipGraph stream capture?  Each case in its own subprocess (a failed capture may abort)."""
import os
import subprocess
import sys

CASES = {
    "kernel_matrix": "K = ops.kernel_matrix(Z, None, variance=1.0, lengthscales=1.0, diag_add=1.0)",
    "potrf_128": "T = A128.clone(); ops.potrf_(T, 128)",
    "potrf_2048": "T = A2048.clone(); ops.potrf_(T, 2048)",
    "potrf_2048_extra": "T = torch.cat([A2048, E]); ops.potrf_(T, 2048)",
}
BODY = '''
import os, sys, faulthandler
faulthandler.enable()
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
sys.path.insert(0, {root!r})
import numpy as np, torch
from gpflow_amd import ops
rng = np.random.default_rng(0)
Z = ops.to_device(rng.normal(size=(2048, 8)))
A2048 = ops.kernel_matrix(Z, None, variance=1.0, lengthscales=2.0, diag_add=1.0)
A128 = A2048[:128, :128].contiguous()
E = ops.to_device(rng.normal(size=(4096, 2048)))
def f():
    {stmt}
for _ in range(3): f()
torch.cuda.synchronize()
s = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    f(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        f()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
print("CAPTURE+REPLAY OK")
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name, stmt in CASES.items():
    r = subprocess.run([sys.executable, "-c", BODY.format(root=root, stmt=stmt)], capture_output=True, text=True, timeout=150,
                       env=dict(os.environ, AMD_LOG_LEVEL="1"))
    tail = (r.stdout + r.stderr).strip().splitlines()[-6:]
    print(f"== {name}: rc={r.returncode}")
    for line in tail:
        print("   ", line[:300])
