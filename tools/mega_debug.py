"""Same-process A/B of the fused SVGP driver: product library (multi-launch route) against the experimental library with
GPK_MEGA=1 (single-launch step kernel, mega.hip) on identical inputs.  Prints the two scalars, the worst differences of the
factor L and of A^T block by block (to localise a wrong tile), and event timings.

    GPK_MEGA=1 [GPK_MEGA_PROTO=1] python tools/mega_debug.py [--sizes m,rows,P ...] [--reps N]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpflow_amd import _lib, ops  # noqa: E402


def load_exp():
    lib = C.CDLL(os.path.join(ROOT, "gpflow_amd", "libgpk_exp.so"))
    lib.gpk_exp_svgp_flags_offset.restype = C.c_long
    lib.gpk_exp_svgp_flags_offset.argtypes = [C.c_int, C.c_int, C.c_int]
    for name, (res, args) in _lib._SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def call(lib, Z, X, Y, q_mu, q_sqrt, ls, ws, out, info):
    m, d = Z.shape
    rows, P = X.shape[0], q_mu.shape[1]
    lsh, ard = ops._ls_host(ls, d)
    rc = lib.gpk_svgp_elbo_shard(torch.cuda.current_stream().cuda_stream, 0, Z.data_ptr(), m, d, X.data_ptr(), Y.data_ptr(), rows, d, P,
                                 d, P, lsh, ard, 1.0, 0.1, 1e-6, 0.0, q_mu.data_ptr(), q_sqrt.data_ptr(), 0, 1, out.data_ptr(),
                                 info.data_ptr(), ws.data_ptr(), ws.numel() * 8)
    if rc:
        raise RuntimeError(f"gpk_svgp_elbo_shard rc={rc}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", nargs="*", default=["128,32,1", "256,64,1", "256,100,2", "512,1000,2", "1024,8192,1", "2048,8192,1",
                                                   "2048,1024,1", "1024,8192,4"])
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    prod, exp = _lib.load(), load_exp()
    print("product:", prod.gpk_version().decode(), "| exp:", exp.gpk_version().decode(), "| GPK_MEGA =", os.environ.get("GPK_MEGA"),
          "PROTO =", os.environ.get("GPK_MEGA_PROTO"))
    dev = ops.device()
    warm = torch.eye(256, dtype=torch.float64, device=dev)
    ops.potrf_(warm, 256)
    torch.cuda.synchronize()
    for spec in args.sizes:
        m, rows, P = (int(v) for v in spec.split(","))
        d = 8
        rng = np.random.default_rng(m + rows)
        Xh = rng.standard_normal((rows + m, d))
        Yh = np.sin(Xh.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((rows + m, P))
        Zh = Xh[:m] + 0.01 * rng.standard_normal((m, d))
        q_mu = ops.to_device(0.1 * rng.standard_normal((m, P)))
        q_sqrt = ops.to_device(np.tril(0.05 * rng.standard_normal((P, m, m))) + 0.5 * np.eye(m))
        Z, X, Y = ops.to_device(Zh), ops.to_device(Xh[m:]), ops.to_device(Yh[m:])
        ls = np.sqrt(d) * (0.8 + 0.05 * np.arange(d))
        nbytes = int(prod.gpk_svgp_elbo_workspace_bytes(m, rows, d, P, 0, 1))
        assert nbytes == int(exp.gpk_svgp_elbo_workspace_bytes(m, rows, d, P, 0, 1))
        res = {}
        for name, lib in (("prod", prod), ("mega", exp)):
            ws = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=dev)
            out = torch.full((2,), float("nan"), dtype=torch.float64, device=dev)
            info = torch.zeros(1, dtype=torch.int32, device=dev)
            t0 = time.perf_counter()
            call(lib, Z, X, Y, q_mu, q_sqrt, ls, ws, out, info)
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            ld = (m + 7) // 8 * 8
            T = ws[:(m + rows) * ld].reshape(m + rows, ld).cpu().numpy().copy()
            o, inf = out.cpu().numpy().copy(), int(info.cpu()[0])
            if name == "mega" and os.environ.get("GPK_MEGA_TRACE"):
                lib.gpk_exp_mega_trace_dump()
            if name == "mega":
                nbp = m // 128
                off = int(lib.gpk_exp_svgp_flags_offset(m, rows, P)) // 8
                nflag = nbp * nbp + 6 * nbp + 8
                st = ws[off + (nflag + 1) // 2: off + (nflag + 1) // 2 + 2 * nbp].cpu().numpy().view(np.int64).reshape(nbp, 2)
                t00 = st[0, 0]
                print("  leaf start / duration (us):", " ".join(f"{(a - t00) / 100:.0f}/{(b - a) / 100:.0f}" for a, b in st))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if inf == 0:
                for _ in range(3):
                    call(lib, Z, X, Y, q_mu, q_sqrt, ls, ws, out, info)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.reps):
                    call(lib, Z, X, Y, q_mu, q_sqrt, ls, ws, out, info)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                o2 = out.cpu().numpy().copy()
            else:
                ms, o2 = float("nan"), o
            res[name] = (T, o, inf, ms, first, o2)
            del ws
        (Tp, op_, ip, msp, _, _), (Tm, om, im, msm, firstm, om2) = res["prod"], res["mega"]
        rel = abs(om[0] - op_[0]) / abs(op_[0])
        print(f"m={m} rows={rows} P={P}: prod out={op_} info={ip} {msp:.3f} ms | mega out={om} info={im} {msm:.3f} ms (first call "
              f"{firstm * 1e3:.1f} ms) | rel diff data term {rel:.2e}, KL {abs(om[1] - op_[1]):.1e}, repeat-identical {np.array_equal(om, om2)}")
        if not (rel < 1e-9) or im != ip:
            nb = m // 128
            L_p, L_m = np.tril(Tp[:m, :m]), np.tril(Tm[:m, :m])
            print("  worst |dL| per 128-block (rows i, cols j):")
            for i in range(nb):
                print("   ", " ".join(f"{np.abs(L_p[128*i:128*i+128, 128*j:128*j+128] - L_m[128*i:128*i+128, 128*j:128*j+128]).max():8.1e}"
                                      for j in range(i + 1)))
            A_p, A_m = Tp[m:, :m], Tm[m:, :m]
            print("  worst |dA^T| per column block:", " ".join(f"{np.abs(A_p[:, 128*j:128*j+128] - A_m[:, 128*j:128*j+128]).max():8.1e}"
                                                              for j in range(nb)))
            rb = np.abs(A_p - A_m).max(axis=1)
            print("  worst |dA^T| per 32-row block (first 16):", " ".join(f"{rb[32*k:32*k+32].max():8.1e}" for k in range(min(16, (rows + 31) // 32))))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
