"""Lane-level NumPy model of the round-6 leaf (gpflow_amd/csrc/leaf2_device.h): the 128 x 128 diagonal-block Cholesky +
inverse with the pivot wave running free of workgroup barriers on its critical data (the next diagonal tile and the tile below
it stay in its registers) and seven helper waves doing everything else in barrier-separated phases.

Two things are checked here, on the CPU, before the HIP code is trusted:
  * the index algebra of the in-register panel steps (v_mfma_f64_16x16x4 operand / result layouts, the look-ahead tile held
    TRANSPOSED in the MFMA C/D layout, the forward-substitution form of the 4 x 4 inverse);
  * the helpers' task schedule: within one phase (between two workgroup barriers) no tile written by one wave is read or written
    by another.
Run:  python tools/leaf2_model.py        (prints the residuals for nb = 128 and a ragged nb; exits non-zero on a hazard)
"""
import sys

import numpy as np

NB, SB = 128, 16
LANES = np.arange(64)
C, G = LANES & 15, LANES >> 4


def mfma4(a, b, c):
    """v_mfma_f64_16x16x4_f64: D[m][n] = C[m][n] + sum_k A[m][k] B[n][k];  A[m][k] = a[lane m + 16 k], B[n][k] = b[lane n + 16 k],
    D[m][n] in register m // 4 of lane n + 16 (m % 4)."""
    A = a.reshape(4, 16).T  # [m, k]
    B = b.reshape(4, 16).T  # [n, k]
    D = A @ B.T             # [m, n]
    out = np.array(c, dtype=float, copy=True)
    for e in range(4):
        for g in range(4):
            out[16 * g:16 * g + 16, e] += D[4 * e + g, :]
    return out


def sym_tile_regs(T):
    """d[e] of lane (c, g) = T[g + 4 e][c] (T symmetric)."""
    d = np.zeros((64, 4))
    for e in range(4):
        d[:, e] = T[G + 4 * e, C]
    return d


def rowmajor_tile_regs(T):
    """a[e] of lane (c, g) = T[c][4 e + g]  (the transposed tile in the C/D layout)."""
    a = np.zeros((64, 4))
    for e in range(4):
        a[:, e] = T[C, 4 * e + G]
    return a


def regs_to_rowmajor(a):
    T = np.zeros((16, 16))
    for e in range(4):
        T[C, 4 * e + G] = a[:, e]
    return T


def regs_to_dlayout(d):
    T = np.zeros((16, 16))
    for e in range(4):
        T[G + 4 * e, C] = d[:, e]
    return T


def panel_pivot(P, d, x, ind):
    pick = lambda a, b: d[4 * P + b + 16 * a, P]
    s00, s10, s20, s30 = pick(0, 0), pick(1, 0), pick(2, 0), pick(3, 0)
    s11, s21, s31 = pick(1, 1), pick(2, 1), pick(3, 1)
    s22, s32, s33 = pick(2, 2), pick(3, 2), pick(3, 3)
    r0 = 1 / np.sqrt(s00)
    l10, l20, l30 = s10 * r0, s20 * r0, s30 * r0
    r1 = 1 / np.sqrt(s11 - l10 * l10)
    l21 = (s21 - l20 * l10) * r1
    l31 = (s31 - l30 * l10) * r1
    r2 = 1 / np.sqrt(s22 - l20 * l20 - l21 * l21)
    l32 = (s32 - l30 * l20 - l31 * l21) * r2
    r3 = 1 / np.sqrt(s33 - l30 * l30 - l31 * l31 - l32 * l32)
    # forward substitution of L4 y = e_g, one column of Y per lane group g (ind[k] = 1 in lanes g == k, c < 4)
    y0 = r0 * ind[0]
    y1 = r1 * (ind[1] - l10 * y0)
    y2 = r2 * (ind[2] - l20 * y0 - l21 * y1)
    y3 = r3 * (ind[3] - l30 * y0 - l31 * y1 - l32 * y2)
    yop = np.where(C == 1, y1, y0)
    yop = np.where(C == 2, y2, yop)
    yop = np.where(C == 3, y3, yop)
    zero = np.zeros((64, 4))
    t = mfma4(yop, d[:, P], zero)
    lp = np.where(C >= 4 * P + G, t[:, 0], 0.0)
    if P < 3:
        d = mfma4(-lp, lp, d)
    u = mfma4(yop, x[:, P], zero)
    xp = u[:, 0]
    if P < 3:
        x = mfma4(-lp, xp, x)
    d[:, P] = lp
    x[:, P] = xp
    return d, x, yop, lp


def panel_ahead(P, a2, s2, yop, lp):
    zero = np.zeros((64, 4))
    t2 = mfma4(yop, a2[:, P], zero)
    lp2 = t2[:, 0]
    if P < 3:
        a2 = mfma4(-lp, lp2, a2)
    s2 = mfma4(-lp2, lp2, s2)
    a2[:, P] = lp2
    return a2, s2


class Hazard(Exception):
    pass


class Phase:
    """Read / write sets of one barrier-to-barrier phase."""

    def __init__(self, name):
        self.name, self.acc = name, []

    def touch(self, wave, reads, writes):
        for (w2, r2, wr2) in self.acc:
            if w2 == wave:
                continue
            for t in writes:
                if t in r2 or t in wr2:
                    raise Hazard(f"{self.name}: wave {wave} writes {t}, wave {w2} uses it")
            for t in reads:
                if t in wr2:
                    raise Hazard(f"{self.name}: wave {wave} reads {t}, wave {w2} writes it")
        self.acc.append((wave, set(reads), set(writes)))


def gamma_tasks(k, n8):
    """Helper tasks of window k between B_k and A_{k+1}, LEFT-LOOKING: the three tiles of row k+3 and the tiles of column block k+1
    below them, each = A(i,j) - sum_{t<=k} L(i,t) L(j,t)^T in one go; row block k+1 of T as multi-term products; then the stores
    of the chain's tiles of column block k."""
    t = []
    if k + 3 < n8:
        for j in range(k + 1, k + 4):
            t.append(("ll", k + 3, j))
    for i in range(k + 4, n8):
        t.append(("ll", i, k + 1))
    if k + 1 < n8:
        for j in range(k + 1):
            t.append(("xrow", k + 1, j))
    for u in range(3):
        if k + u < n8:
            t.append(("stL", k + u, k))
    t.append(("stX", k, k))
    return t


def alpha_tasks(k, n8):
    """between A_k and B_k: L(i,k) for the rows below the chain's two, last product of row block k of the inverse"""
    t = [("B", i, k) for i in range(k + 3, n8)]
    t += [("xfin", k, j) for j in range(k)]
    return t


def panel_pivot3(P, d, ind):
    """pivot wave: as panel_pivot without the X rows; returns what it publishes (yop, lp)"""
    x = np.zeros((64, 4))
    d, _, yop, lp = panel_pivot(P, d, x, ind)
    return d, yop, lp


def panel_trail(P, x, a2, s2, yop, lp, ahead):
    zero = np.zeros((64, 4))
    u = mfma4(yop, x[:, P], zero)
    xp = u[:, 0]
    if ahead:
        t2 = mfma4(yop, a2[:, P], zero)
        lp2 = t2[:, 0]
        s2 = mfma4(-lp2, lp2, s2)
    if P < 3:
        x = mfma4(-lp, xp, x)
        if ahead:
            a2 = mfma4(-lp, lp2, a2)
    x[:, P] = xp
    if ahead:
        a2[:, P] = lp2
    return x, a2, s2


def leaf2(A, nb):
    """Round-6 leaf, final form: two chain waves alternate as pivot wave (PW) and trailing wave (TW); six helpers."""
    n8 = (nb + SB - 1) // SB
    S = np.full((NB, NB), np.nan)
    # LDS image: the helpers load rows >= 32 only; tiles (0,0), (1,0), (1,1) go from global memory straight to the chain waves
    S[32:, :] = 0.0
    if nb > 32:
        S[32:nb, :nb] = np.tril(A[:nb, :nb])[32:, :]
    for i in range(max(nb, 32), NB):
        S[i, i] = 1.0
    Apad = np.eye(NB)
    Apad[:nb, :nb] = A[:nb, :nb]
    Xt, Xd = {}, {}
    Lout = np.zeros((NB, NB))
    inv = np.zeros((NB, NB))
    for i in range(16 * n8, NB):
        inv[i, i] = 1.0
    tile = lambda i, j: S[16 * i:16 * i + 16, 16 * j:16 * j + 16]
    gtile = lambda i, j: Apad[16 * i:16 * i + 16, 16 * j:16 * j + 16]
    ind = [np.where((G == k) & (C < 4), 1.0, 0.0) for k in range(4)]

    def run_task(ph, wave, task, k):
        kind, i, j = task
        if kind == "B":
            ph.touch(wave, {("L", i, k), ("Xd", k)}, {("L", i, k)})
            tile(i, k)[:] = tile(i, k) @ Xd[k].T
            Lout[16 * i:16 * i + 16, 16 * k:16 * k + 16] = tile(i, k)
        elif kind == "xfin":
            ph.touch(wave, {("X", i, j), ("Xd", i)}, {("X", i, j)})
            Xt[(i, j)] = -Xd[i] @ Xt[(i, j)]
            inv[16 * i:16 * i + 16, 16 * j:16 * j + 16] = Xt[(i, j)]
        elif kind == "ll":    # tile (i,j) = A(i,j) - sum_{t<=k} L(i,t) L(j,t)^T
            ph.touch(wave, {("L", i, j)} | {("L", i, t) for t in range(k + 1)} | {("L", j, t) for t in range(k + 1)}, {("L", i, j)})
            acc = tile(i, j).copy()
            for t in range(k + 1):
                acc -= tile(i, t) @ tile(j, t).T
            tile(i, j)[:] = acc
        elif kind == "xrow":  # T(i,j) = sum_{t=j}^{i-1} L(i,t) X(t,j)
            reads = {("L", i, t) for t in range(j, i)} | {("Xd", j)} | {("X", t, j) for t in range(j + 1, i)}
            ph.touch(wave, reads, {("X", i, j)})
            acc = tile(i, j) @ Xd[j]
            for t in range(j + 1, i):
                acc = acc + tile(i, t) @ Xt[(t, j)]
            Xt[(i, j)] = acc
        elif kind == "stL":
            ph.touch(wave, {("L", i, j)}, set())
            Lout[16 * i:16 * i + 16, 16 * j:16 * j + 16] = tile(i, j) if i != j else np.tril(tile(i, j))
        elif kind == "stX":
            ph.touch(wave, {("Xd", i)} if i == j else {("X", i, j)}, set())
            inv[16 * i:16 * i + 16, 16 * j:16 * j + 16] = Xd[i] if i == j else Xt[(i, j)]

    def ident_x():
        x = np.zeros((64, 4))
        for e in range(4):
            x[:, e] = np.where(4 * e + G == C, 1.0, 0.0)
        return x

    symm = lambda T: np.tril(T) + np.tril(T, -1).T
    # start: PW(0) has tile (0,0), TW(0) the look-ahead pair, all from global memory
    d = sym_tile_regs(symm(gtile(0, 0)))
    a2 = rowmajor_tile_regs(gtile(1, 0)) if n8 > 1 else None
    s2 = sym_tile_regs(symm(gtile(1, 1))) if n8 > 1 else None
    for k in range(n8):
        ahead = k + 1 < n8
        # phases seen from the chain: tile k overlaps  gamma(k-2) | A_{k-1} | alpha(k-1) | B_{k-1} | gamma(k-1) | A_k
        # -- PW: panel 0 before A_{k-1}; TW(k) forms L(k+1,k-1) between A_{k-1} and B_{k-1}
        pub = []
        d, yop, lp = panel_pivot3(0, d, ind); pub.append((yop, lp))
        ph = Phase(f"alpha({k - 1})")
        if k >= 1:
            for n, t in enumerate(alpha_tasks(k - 1, n8)):
                run_task(ph, 10 + n, t, k - 1)
            if ahead:
                # Lr = L(k+1,k-1) in fragment layout: D[m][n] = sum_q X[m][q] A(k+1,k-1)[n][q]
                ph.touch(1, {("Xd", k - 1), ("L", k + 1, k - 1)}, {("L", k + 1, k - 1)})
                fa = rowmajor_tile_regs(Xd[k - 1])          # fa[kk] of lane (r, kq) = X[r][4kk+kq]
                fb = rowmajor_tile_regs(tile(k + 1, k - 1))
                Lr = np.zeros((64, 4))
                for kk in range(4):
                    Lr = mfma4(fa[:, kk], fb[:, kk], Lr)
                tile(k + 1, k - 1)[:] = regs_to_rowmajor(Lr)   # Lr[e] of lane (c, g) = L(k+1,k-1)[c][4e+g]
        d, yop, lp = panel_pivot3(1, d, ind); pub.append((yop, lp))
        # -- B_{k-1}: TW forms the look-ahead pair, then follows the published panels; helpers gamma(k-1)
        ph = Phase(f"gamma({k - 1})")
        if k >= 1:
            for n, t in enumerate(gamma_tasks(k - 1, n8)):
                run_task(ph, 10 + n, t, k - 1)
            # early terms (t < k) of window k's tasks run in this phase too, on whichever helper gets there first: reads only
            for n, (kind, i, j) in enumerate(gamma_tasks(k, n8) if k + 1 < n8 else []):
                if kind == "ll":
                    ph.touch(50 + n, {("L", i, t) for t in range(k)} | {("L", j, t) for t in range(k)}, set())
                elif kind == "xrow":
                    if j < k:
                        ph.touch(50 + n, {("L", i, t) for t in range(j, k)} | {("Xd", j)} | {("X", t, j) for t in range(j + 1, k)}, set())
            if ahead:
                ph.touch(1, {("L", k, k - 1), ("L", k + 1, k), ("L", k + 1, k + 1)}, {("Xd", k), ("L", k + 1, k)})
                fl = rowmajor_tile_regs(tile(k, k - 1))
                a2 = rowmajor_tile_regs(tile(k + 1, k))
                s2 = sym_tile_regs(symm(tile(k + 1, k + 1)))
                for kk in range(4):
                    a2 = mfma4(-fl[:, kk], Lr[:, kk], a2)
                    s2 = mfma4(-Lr[:, kk], Lr[:, kk], s2)
            else:
                ph.touch(1, set(), {("Xd", k)})
        ph.touch(0, set(), {("L", k, k)})
        d, yop, lp = panel_pivot3(2, d, ind); pub.append((yop, lp))
        d, yop, lp = panel_pivot3(3, d, ind); pub.append((yop, lp))
        x = ident_x()
        for P in range(4):
            x, a2, s2 = panel_trail(P, x, a2, s2, pub[P][0], pub[P][1], ahead)
        tile(k, k)[:] = np.tril(regs_to_rowmajor(d))
        Xd[k] = regs_to_dlayout(x)
        if ahead:
            tile(k + 1, k)[:] = regs_to_rowmajor(a2)
            d = s2
    k = n8 - 1
    ph = Phase("tail alpha")
    for n, t in enumerate(alpha_tasks(k, n8)):
        run_task(ph, 10 + n, t, k)
    ph = Phase("tail stores")
    for n, t in enumerate(gamma_tasks(k, n8)):
        run_task(ph, 10 + n, t, k)
    return Lout, inv


def check(nb, seed=0):
    rng = np.random.default_rng(seed)
    B = rng.normal(size=(nb, nb))
    A = B @ B.T + nb * np.eye(nb)
    L, inv = leaf2(A, nb)
    Lref = np.linalg.cholesky(A)
    e1 = np.abs(L[:nb, :nb] - Lref).max() / np.abs(Lref).max()
    Xref = np.eye(NB)
    Xref[:nb, :nb] = np.linalg.inv(Lref)
    n8 = (nb + SB - 1) // SB
    e2 = np.abs(inv - Xref).max() / np.abs(Xref).max()
    print(f"nb = {nb:3d} (tiles {n8}): |L - chol| = {e1:.2e}   |inv - L^-1| = {e2:.2e}")
    return max(e1, e2)


if __name__ == "__main__":
    worst = 0.0
    try:
        for nb in (128, 100, 64, 17, 16, 5, 33, 112):
            worst = max(worst, check(nb))
    except Hazard as h:
        print("HAZARD:", h)
        sys.exit(2)
    print("worst", worst)
    sys.exit(0 if worst < 1e-11 else 1)
