"""The three triangular x triangular products of the Cholesky adjoint (gradients.cholesky_adjoint) at M = 2048, alone on the chip:
unsplit against K chunks in the batch dimension (ops.gemm_nt k_split) + the combine pass."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpflow_amd import ops

def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(0)
U = ops.to_device(np.triu(rng.normal(size=(n, n))))      # upper: L^T, L^-T
Lw = ops.to_device(np.tril(rng.normal(size=(n, n))))     # lower: Phi
for name, A, B, a_tri, b_tri in (("upper x upper^T (T1, S)", U, U, 1, 1), ("lower x upper^T (Y)", Lw, U, 2, 1)):
    ref = ops.gemm_nt(A, B, b_tri=b_tri, a_tri=a_tri)
    print(f"{name}: unsplit {timed(lambda: ops.gemm_nt(A, B, b_tri=b_tri, a_tri=a_tri)):.1f} us", flush=True)
    for chunks in (2, 4, 8):
        kc = n // chunks
        A3 = torch.as_strided(A, (chunks, n, kc), (kc, n, 1)); B3 = torch.as_strided(B, (chunks, n, kc), (kc, n, 1))
        parts = ops.gemm_nt(A3, B3, b_tri=b_tri, a_tri=a_tri, k_split=True)
        err = float((ops.combine_parts(parts) - ref).abs().max() / ref.abs().max())
        tg = timed(lambda: ops.gemm_nt(A3, B3, b_tri=b_tri, a_tri=a_tri, k_split=True, C=parts))
        tc = timed(lambda: ops.combine_parts(parts))
        print(f"   {chunks} chunks: gemm {tg:.1f} us + combine {tc:.1f} us   (max rel diff {err:.1e})", flush=True)

# controls: dense launches of the same tile shape
kc = n // 4
A3 = torch.as_strided(U, (4, n, kc), (kc, n, 1)); B3 = torch.as_strided(U, (4, n, kc), (kc, n, 1))
C3 = torch.empty((4, n, n), dtype=torch.float64, device=U.device)
print(f"dense batch of 4, K = {kc}: {timed(lambda: ops.gemm_nt(A3, B3, C=C3)):.1f} us")
print(f"dense single, K = {kc}: {timed(lambda: ops.gemm_nt(A3[3], B3[3], C=C3[3])):.1f} us")
print(f"dense single, K = {n}: {timed(lambda: ops.gemm_nt(U, U, C=C3[0])):.1f} us")
for z in range(4):
    # chunk z alone with the structure of the unsplit product emulated by sub-matrix sizes: rows/cols < 128 * 4 (z + 1)
    mz = 512 * (z + 1)
    print(f"chunk {z} alone as a dense [{mz} x {mz}] x {kc} launch: {timed(lambda: ops.gemm_nt(A3[z][:mz], B3[z][:mz], C=C3[z][:mz, :mz])):.1f} us")
