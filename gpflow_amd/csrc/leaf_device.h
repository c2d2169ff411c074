// Device code of the 128 x 128 diagonal-block factorisation (leaf): shared by the stand-alone leaf kernel (leaf.hip) and
// the single-launch SVGP step kernel (mega.hip), which runs the same leaf inside a persistent workgroup.
// (Moved verbatim out of leaf.hip in round 4; the description of the algorithm is at the top of leaf.hip.)
#pragma once
#include "gpk_internal.h"

namespace gpk_leaf {

constexpr int NB = GPK_NB;
constexpr int LD = NB + 2;     // 130: A-layout fragment reads hit 64 distinct banks
constexpr int SB = 16;         // sub-block
constexpr int NSB = NB / SB;   // 8
constexpr int XLD = SB + 1;    // row stride of the dense diagonal tiles of X
constexpr int NT = 512;        // 8 waves
constexpr int NW = NT / 64;
constexpr int XD_OFF = NB * LD;                  // doubles: [NSB][SB][XLD]
constexpr int LDS_DOUBLES = XD_OFF + NSB * SB * XLD;
constexpr size_t LEAF_LDS = (size_t)LDS_DOUBLES * sizeof(double);

__device__ __forceinline__ double readlane_d(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

// 1/sqrt(p): v_rsq_f64 (2^-23 relative) + ONE third-order step  y (1 + e/2 + 3e^2/8),  e = 1 - p y^2
// (error 5/16 e^3 ~ 2^-70).  Five dependent fp64 ops instead of the seven of two Newton steps: the
// dependent-issue latency of fp64 VALU ops (~38 cycles) times the 128 pivots IS the leaf's critical path.
__device__ __forceinline__ double rsqrt_nr(double p) {
  const double y = __builtin_amdgcn_rsq(p);
  const double t = p * y;
  const double e = fma(-t, y, 1.0);
  const double q = fma(0.375, e, 0.5);
  const double s = y * e;
  return fma(s, q, y);
}

__device__ __forceinline__ d4 mfma4(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---- 16x16 diagonal tile: Cholesky + inverse in the registers of one wave ---------------------------------
// c = lane & 15, g = lane >> 4.  On entry (FACTORED = false) d[e] = S[g+4e][c] of the symmetric tile; on
// entry (FACTORED = true) d[e] = L[c][4e+g] (zero above the diagonal).  On exit d[e] = L[c][4e+g] and
// x[e] = X[4e+g][c]  (X = L^-1, exact zeros above the diagonal).
template <bool FACTORED, int P>
__device__ __forceinline__ void diag_panel(d4& d, d4& x, int c, int g, int lane, int& bad_col, int col0) {
  if constexpr (P < 4) {
    // 4x4 pivot block  s[a][b] = S[4P+a][4P+b]  (a >= b), wave-uniform
    auto pick = [&](int a, int b) -> double {
      // !FACTORED: d[P] of lane (c = 4P+b, g = a);  FACTORED: d[P] of lane (c = 4P+a, g = b)
      return FACTORED ? readlane_d(d[P], 4 * P + a + 16 * b) : readlane_d(d[P], 4 * P + b + 16 * a);
    };
    const double s00 = pick(0, 0), s10 = pick(1, 0), s20 = pick(2, 0), s30 = pick(3, 0);
    const double s11 = pick(1, 1), s21 = pick(2, 1), s31 = pick(3, 1);
    const double s22 = pick(2, 2), s32 = pick(3, 2), s33 = pick(3, 3);
    double yop;
    const int slot = (c < 4 && g <= c) ? c * 4 + g : -1;  // lane (m = c, k = g) holds Y[m][k] of the A-operand
#ifdef GPK_LEAF_FRACTION_FREE
    constexpr bool kRecurrence = FACTORED;   // (A/B only, `make ffleaf`: the fraction-free pivot block below)
#else
    constexpr bool kRecurrence = true;
#endif
    if constexpr (kRecurrence) {
      double l10, l20, l30, l21, l31, l32, r0, r1, r2, r3;
      if constexpr (FACTORED) {
        l10 = s10; l20 = s20; l30 = s30; l21 = s21; l31 = s31; l32 = s32;
        r0 = 1.0 / s00; r1 = 1.0 / s11; r2 = 1.0 / s22; r3 = 1.0 / s33;
      } else {
        r0 = rsqrt_nr(s00);
        l10 = s10 * r0; l20 = s20 * r0; l30 = s30 * r0;
        const double p1 = fma(-l10, l10, s11);
        r1 = rsqrt_nr(p1);
        l21 = fma(-l20, l10, s21) * r1;
        l31 = fma(-l30, l10, s31) * r1;
        const double p2 = fma(-l21, l21, fma(-l20, l20, s22));
        r2 = rsqrt_nr(p2);
        l32 = fma(-l31, l21, fma(-l30, l20, s32)) * r2;
        const double p3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, s33)));
        r3 = rsqrt_nr(p3);
        int idx = -1;
        idx = !(p3 > 0.0) ? 3 : idx;
        idx = !(p2 > 0.0) ? 2 : idx;
        idx = !(p1 > 0.0) ? 1 : idx;
        idx = !(s00 > 0.0) ? 0 : idx;
        bad_col = (bad_col < 0 && idx >= 0) ? col0 + 4 * P + idx : bad_col;
      }
      // Y = inv(L4), lower triangular
      const double y10 = -r1 * (l10 * r0);
      const double y21 = -r2 * (l21 * r1);
      const double y32 = -r3 * (l32 * r2);
      const double y20 = -r2 * fma(l21, y10, l20 * r0);
      const double y31 = -r3 * fma(l32, y21, l31 * r1);
      const double y30 = -r3 * fma(l32, y20, fma(l31, y10, l30 * r0));
      // flat select chain on the per-lane slot index (no divergent control flow: every Y value is wave-uniform)
      yop = 0.0;
      yop = (slot == 0) ? r0 : yop;
      yop = (slot == 4) ? y10 : yop;
      yop = (slot == 5) ? r1 : yop;
      yop = (slot == 8) ? y20 : yop;
      yop = (slot == 9) ? y21 : yop;
      yop = (slot == 10) ? r2 : yop;
      yop = (slot == 12) ? y30 : yop;
      yop = (slot == 13) ? y31 : yop;
      yop = (slot == 14) ? y32 : yop;
      yop = (slot == 15) ? r3 : yop;
    } else {
      // ROUND-5 EXPERIMENT, NOT THE PRODUCT PATH (compiled only with -DGPK_LEAF_FRACTION_FREE): Y = inv(chol(S4)) without a
      // square root or a division on the dependent path.  The recurrence above (pivot -> rsqrt -> scale the column -> next
      // pivot) is ~36 DEPENDENT fp64 ops per 4-column panel; the form below needs 15 levels.  Measured on the chip it is 0.5 - 1.5 %
      // SLOWER at step level (profiles/r05_ab_leaf_fraction_free.log: factor phase 23 - 25 us either way): wave 0 is bound by the
      // ISSUE of ~100 wave-uniform VALU instructions per panel (7 - 9 cycles each), not by their dependent latency, and this form
      // issues ~15 more of them.  Kept as the record of that finding.  Fraction-free elimination:
      //   t_ij = s00 s_ij - s_i0 s_j0,   u_ij = t11 t_ij - t_i1 t_j1,   w33 = u22 u33 - u32^2
      // are the Schur complements scaled by the previous pivots (s00 = p0, t11 = p0 p1, u22 = p0^2 p1 p2, w33 = p0^4 p1^2 p2 p3),
      // the same row operations applied to the identity give the rows N_k of the unit-lower inverse times those scales, and
      //   Y[k][:] = (q0 q1 ... qk) N_k   with  q0 = rsqrt(s00), q1 = rsqrt(t11), q2 = rsqrt(u22), q3 = rsqrt(w33):
      // six dependent ops to the last scaled pivot, the four rsqrt refinements run beside each other, 15 levels in all.
      // Same backward error as the recurrence (tools/leaf_ff_check.py: |Y S Y^T - I| equal to within a factor 1.5 for
      // condition numbers 1e1 ... 1e12).  Range: intermediate magnitudes reach pivot^8, so entries beyond ~1e+-35 over/underflow
      // -- and are then REPORTED as a non-positive pivot, never silently accepted (NaN / 0 fail the `> 0` tests below).
      const double t11 = fma(s00, s11, -(s10 * s10));
      const double t21 = fma(s00, s21, -(s20 * s10));
      const double t31 = fma(s00, s31, -(s30 * s10));
      const double t22 = fma(s00, s22, -(s20 * s20));
      const double t32 = fma(s00, s32, -(s30 * s20));
      const double t33 = fma(s00, s33, -(s30 * s30));
      const double u22 = fma(t11, t22, -(t21 * t21));
      const double u32 = fma(t11, t32, -(t31 * t21));
      const double u33 = fma(t11, t33, -(t31 * t31));
      const double w33 = fma(u22, u33, -(u32 * u32));
      // the four rsqrt refinements (rsqrt_nr, written out) level by level, so that the in-order issue of the wave sees four
      // independent ops per level instead of four serial six-op chains; the rows of the scaled unit-lower inverse fill the slots
      const double y0 = __builtin_amdgcn_rsq(s00), y1 = __builtin_amdgcn_rsq(t11), y2 = __builtin_amdgcn_rsq(u22),
                   y3 = __builtin_amdgcn_rsq(w33);
      const double a = t11 * s00;
      const double f0 = s00 * y0, f1 = t11 * y1, f2 = u22 * y2, f3 = w33 * y3;
      const double n20 = fma(t21, s10, -(t11 * s20)), n21 = -(t21 * s00);
      const double e0 = fma(-f0, y0, 1.0), e1 = fma(-f1, y1, 1.0), e2 = fma(-f2, y2, 1.0), e3 = fma(-f3, y3, 1.0);
      const double m30 = fma(t31, s10, -(t11 * s30)), m31 = -(t31 * s00);
      const double g0 = fma(0.375, e0, 0.5), g1 = fma(0.375, e1, 0.5), g2 = fma(0.375, e2, 0.5), g3 = fma(0.375, e3, 0.5);
      const double h0 = y0 * e0, h1 = y1 * e1, h2 = y2 * e2, h3 = y3 * e3;
      const double n30 = fma(u22, m30, -(u32 * n20)), n31 = fma(u22, m31, -(u32 * n21)), n32 = -(u32 * a), n33 = u22 * a;
      const double q0 = fma(h0, g0, y0), q1 = fma(h1, g1, y1), q2 = fma(h2, g2, y2), q3 = fma(h3, g3, y3);
      // first non-positive pivot of this panel, branch-free (all values are wave-uniform); the scaled pivots have the sign
      // of the true ones as long as every earlier pivot is positive, which is all the "first failure" needs
      int idx = -1;
      idx = !(w33 > 0.0) ? 3 : idx;
      idx = !(u22 > 0.0) ? 2 : idx;
      idx = !(t11 > 0.0) ? 1 : idx;
      idx = !(s00 > 0.0) ? 0 : idx;
      bad_col = (bad_col < 0 && idx >= 0) ? col0 + 4 * P + idx : bad_col;
      // per-lane operand  Y[m][k] = N[m][k] * rho_m:  the N entries are ready early (select chain off the critical path), the
      // cumulative products rho_m = q0 ... qm last -- ONE select and one multiply behind q3
      double nsel = 0.0;
      nsel = (slot == 0) ? 1.0 : nsel;
      nsel = (slot == 4) ? -s10 : nsel;
      nsel = (slot == 5) ? s00 : nsel;
      nsel = (slot == 8) ? n20 : nsel;
      nsel = (slot == 9) ? n21 : nsel;
      nsel = (slot == 10) ? a : nsel;
      nsel = (slot == 12) ? n30 : nsel;
      nsel = (slot == 13) ? n31 : nsel;
      nsel = (slot == 14) ? n32 : nsel;
      nsel = (slot == 15) ? n33 : nsel;
      const double rho1 = q0 * q1, q23 = q2 * q3, rho2 = rho1 * q2, rho3 = rho1 * q23;
      double rsel = q0;
      rsel = (c == 1) ? rho1 : rsel;
      rsel = (c == 2) ? rho2 : rsel;
      rsel = (c == 3) ? rho3 : rsel;
      yop = nsel * rsel;
    }
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    // panel of L:  D[m][n] = sum_k Y[m][k] S[n][4P+k]  ->  reg 0 of lane (n, g) = L[n][4P+g]
    double lp;
    if constexpr (FACTORED) {
      lp = d[P];
    } else {
      const d4 t = mfma4(yop, d[P], zero);
      lp = (c >= 4 * P + g) ? t[0] : 0.0;  // rows above the panel and the upper part of the 4x4 block
    }
    if constexpr (!FACTORED && P < 3) d = mfma4(-lp, lp, d);  // S -= Lp Lp^T  (critical: next pivots)
    // new rows of X:  Xp[m][n] = sum_k Y[m][k] Xtmp[4P+k][n]  ->  reg 0 of lane (n, g) = X[4P+g][n]
    const d4 u = mfma4(yop, x[P], zero);
    const double xp = u[0];
    if constexpr (P < 3) x = mfma4(-lp, xp, x);  // Xtmp[m][:] -= L[m][4P+k] X[4P+k][:]
    d[P] = lp;
    x[P] = xp;
    diag_panel<FACTORED, P + 1>(d, x, c, g, lane, bad_col, col0);
  }
}

template <bool FACTORED>
__device__ __forceinline__ void diag16(double* __restrict__ S, int k, int lane, int& bad_col, int col0) {
  const int c = lane & 15, g = lane >> 4;
  const int kb = k * SB;
  d4 d, x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = g + 4 * e;  // the other index
    if constexpr (FACTORED) {
      d[e] = (c >= j) ? S[(kb + c) * LD + kb + j] : 0.0;           // L[c][4e+g]
    } else {
      d[e] = (j >= c) ? S[(kb + j) * LD + kb + c] : S[(kb + c) * LD + kb + j];  // S[g+4e][c], symmetric
    }
    x[e] = (j == c) ? 1.0 : 0.0;
  }
  diag_panel<FACTORED, 0>(d, x, c, g, lane, bad_col, col0 + kb);
  double* __restrict__ Xd = S + XD_OFF + k * (SB * XLD);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * e + g;
    if (!FACTORED && c >= j) S[(kb + c) * LD + kb + j] = d[e];  // L[c][j]
    Xd[j * XLD + c] = x[e];                                     // X[j][c]
  }
}

// ---- 16x16 tile products on LDS-resident operands --------------------------------------------------------
// A tile reference: element (r, q) lives at S[base + r * rs + q * cs].
struct TRef { int base, rs, cs; };
__device__ __forceinline__ TRef tile_L(int i, int j) { return {i * SB * LD + j * SB, LD, 1}; }       // L_ij[r][q]
__device__ __forceinline__ TRef tile_X(int i, int j) { return {j * SB * LD + i * SB, 1, LD}; }       // X_ij (i>j), stored transposed
__device__ __forceinline__ TRef tile_Xd(int k) { return {XD_OFF + k * SB * XLD, XLD, 1}; }          // X_kk dense
__device__ __forceinline__ TRef tr(TRef t) { return {t.base, t.cs, t.rs}; }                          // transposed view

struct Frag { double a[4], b[4]; };
// operands of  D[m][n] += sum_q A[m][q] * B[n][q]   (NT form; pass tr(B) for a plain product)
__device__ __forceinline__ void frag_load(const double* __restrict__ S, TRef A, TRef B, int lane, Frag& f) {
  const int r = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    f.a[kk] = S[A.base + r * A.rs + (4 * kk + kq) * A.cs];
    f.b[kk] = S[B.base + r * B.rs + (4 * kk + kq) * B.cs];
  }
}
template <bool NEG>
__device__ __forceinline__ d4 frag_mma(const Frag& f, d4 acc) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) acc = mfma4(NEG ? -f.a[kk] : f.a[kk], f.b[kk], acc);
  return acc;
}
__device__ __forceinline__ d4 tile_load(const double* __restrict__ S, TRef C, int lane) {
  const int c = lane & 15, g = lane >> 4;
  d4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = S[C.base + (g + 4 * e) * C.rs + c * C.cs];
  return v;
}
__device__ __forceinline__ void tile_store(double* __restrict__ S, TRef C, int lane, d4 v) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) S[C.base + (g + 4 * e) * C.rs + c * C.cs] = v[e];
}
// second factor already in registers in D layout (T[q][n]: lane (n, g) reg kk = T[4kk+g][n]):
//   D[m][n] += sum_q A[m][q] * T[q][n]
template <bool NEG>
__device__ __forceinline__ d4 reg_mma(const double* __restrict__ S, TRef A, d4 t, int lane, d4 acc) {
  const int r = lane & 15, kq = lane >> 4;
  double a[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) a[kk] = S[A.base + r * A.rs + (4 * kk + kq) * A.cs];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) acc = mfma4(NEG ? -a[kk] : a[kk], t[kk], acc);
  return acc;
}

// linear index u over the lower-triangular tile list  (i, j), j0 <= j <= i, row-major from i = j0
__device__ __forceinline__ void tri_decode(int u, int j0, int& i, int& j) {
  int row = 0;
  while (u > row) { u -= row + 1; ++row; }
  i = j0 + row;
  j = j0 + u;
}

// ---- pieces of the recursive-doubling assembly of X = L^-1 (X21 = -X22 (L21 X11) at block sizes 16, 32, 64) ------
// level 1, node p: tile (2p+1, 2p); one wave, the intermediate T stays in registers
__device__ __forceinline__ void inv_level1(double* __restrict__ S, int p, int lane) {
  Frag f;
  frag_load(S, tile_L(2 * p + 1, 2 * p), tr(tile_Xd(2 * p)), lane, f);            // T = L21 X11
  const d4 t = frag_mma<false>(f, (d4){0.0, 0.0, 0.0, 0.0});
  const d4 r = reg_mma<true>(S, tile_Xd(2 * p + 1), t, lane, (d4){0.0, 0.0, 0.0, 0.0});  // -X22 T
  tile_store(S, tile_X(2 * p + 1, 2 * p), lane, r);
}
// level 2, node q (blocks 4q..4q+3): tile X21(a, b), one wave per (b, a)
__device__ __forceinline__ void inv_level2(double* __restrict__ S, int q, int b, int a, int lane) {
  const int r0 = 4 * q + 2, c0 = 4 * q;  // tile coordinates of the node's L21 / X21 block
  // T(t, b) = sum_{s >= b} L21(t, s) X11(s, b),  t = 0..a   (X11(s,b): s == b diagonal tile, s > b off-diagonal)
  d4 t0 = {0.0, 0.0, 0.0, 0.0}, t1 = {0.0, 0.0, 0.0, 0.0};
  for (int s2 = b; s2 < 2; ++s2) {
    const TRef xs = (s2 == b) ? tile_Xd(c0 + b) : tile_X(c0 + s2, c0 + b);
    Frag f;
    frag_load(S, tile_L(r0, c0 + s2), tr(xs), lane, f);
    t0 = frag_mma<false>(f, t0);
    if (a == 1) {
      frag_load(S, tile_L(r0 + 1, c0 + s2), tr(xs), lane, f);
      t1 = frag_mma<false>(f, t1);
    }
  }
  // X21(a, b) = -sum_{t <= a} X22(a, t) T(t, b)
  d4 r = {0.0, 0.0, 0.0, 0.0};
  if (a == 0) {
    r = reg_mma<true>(S, tile_Xd(r0), t0, lane, r);
  } else {
    r = reg_mma<true>(S, tile_X(r0 + 1, r0), t0, lane, r);
    r = reg_mma<true>(S, tile_Xd(r0 + 1), t1, lane, r);
  }
  tile_store(S, tile_X(r0 + a, c0 + b), lane, r);
}
// level 3, phase 1: T(t, b) = sum_{s=b}^{3} L(4+t, s) X(s, b), parked (transposed, like X) in the X21 region
__device__ __forceinline__ void inv_level3_T(double* __restrict__ S, int t, int b, int lane) {
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int s2 = b; s2 < 4; ++s2) {
    const TRef xs = (s2 == b) ? tile_Xd(b) : tile_X(s2, b);
    Frag f;
    frag_load(S, tile_L(4 + t, s2), tr(xs), lane, f);
    acc = frag_mma<false>(f, acc);
  }
  tile_store(S, tile_X(4 + t, b), lane, acc);
}
// level 3, phase 2: X21(a, b) = -sum_{t=0}^{a} X22(a, t) T(t, b)   (result returned, stored after a barrier)
__device__ __forceinline__ d4 inv_level3_X(const double* __restrict__ S, int a, int b, int lane) {
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int t = 0; t <= a; ++t) {
    const TRef xa = (t == a) ? tile_Xd(4 + a) : tile_X(4 + a, 4 + t);
    Frag f;
    frag_load(S, xa, tr(tile_X(4 + t, b)), lane, f);
    acc = frag_mma<true>(f, acc);
  }
  return acc;
}

// The leaf as a device function (one workgroup of NT threads, LDS block S of LEAF_LDS bytes): used by the
// stand-alone kernel below.
// WT: the results (L block, inverse block) are written with agent-scope write-through stores, so that workgroups on other
// XCDs can read them after a flag hand-off without this workgroup flushing its L2 (mega.hip).
template <bool FACTORED, bool WT = false>
__device__ __forceinline__ void leaf_body(double* __restrict__ S, double* __restrict__ A, long lda, int nb,
                                          double* __restrict__ inv, int* __restrict__ info, int col0,
                                          long long* __restrict__ dbg) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsb = (nb + SB - 1) / SB;
  const long long t_begin = dbg ? wall_clock64() : 0;

  // ---- load: lower triangle of A (identity beyond nb), zero strict upper; all loads issued up front ----
  {
    // thread t owns the column pair (2 (t & 63), +1) of rows (t >> 6) + 8 it
    const int jp = 2 * (tid & 63), r0 = tid >> 6;
    const bool vec = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    constexpr int NIT = NB / NW;
    d2 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = r0 + NW * it;
      v[it] = (d2){0.0, 0.0};
      if (i < nb && jp <= i) {
        const double* src = A + (long)i * lda + jp;
        if (vec && jp + 1 < nb) v[it] = *reinterpret_cast<const d2*>(src);
        else { v[it].x = src[0]; if (jp + 1 < nb) v[it].y = src[1]; }
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = r0 + NW * it;
      d2 w = v[it];
      if (jp > i) w.x = 0.0; else if (i >= nb) w.x = (jp == i) ? 1.0 : 0.0;
      if (jp + 1 > i) w.y = 0.0; else if (i >= nb) w.y = (jp + 1 == i) ? 1.0 : 0.0;
      *reinterpret_cast<d2*>(&S[i * LD + jp]) = w;
    }
    // diagonal tiles of X beyond the factored range are the identity
    for (int e = tid; e < NSB * SB * XLD; e += NT) {
      const int r = (e / XLD) % SB, q = e % XLD;
      S[XD_OFF + e] = (r == q) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  const long long t_loaded = dbg ? wall_clock64() : 0;

  int bad_col = -1;
  const bool overlap_inv = !FACTORED && nsb == NSB;
  if constexpr (FACTORED) {
    if (wave < nsb) diag16<true>(S, wave, lane, bad_col, 0);
    __syncthreads();
  } else {
    if (wave == 0) diag16<false>(S, 0, lane, bad_col, 0);
    __syncthreads();
    for (int k = 0; k < nsb - 1; ++k) {
      // ---- B:  L_ik = A_ik X_kk^T,  i = k+1 .. nsb-1 ---------------------------------------------
      for (int i = k + 1 + wave; i < nsb; i += NW) {
        Frag f;
        frag_load(S, tile_L(i, k), tile_Xd(k), lane, f);
        const d4 r = frag_mma<false>(f, (d4){0.0, 0.0, 0.0, 0.0});
        tile_store(S, tile_L(i, k), lane, r);
      }
      __syncthreads();
      // ---- C:  A_ij -= L_ik L_jk^T ; wave 0: tile (k+1,k+1) then the next diagonal tile ------------
      if (wave == 0) {
        Frag f;
        frag_load(S, tile_L(k + 1, k), tile_L(k + 1, k), lane, f);
        d4 acc = tile_load(S, tile_L(k + 1, k + 1), lane);
        acc = frag_mma<true>(f, acc);
        tile_store(S, tile_L(k + 1, k + 1), lane, acc);
        // the tile was written and is re-read by this wave only (LDS ops of one wave stay ordered)
        diag16<false>(S, k + 1, lane, bad_col, 0);
      } else {
        const int rem = nsb - 1 - k;                 // rows k+1 .. nsb-1
        const int ntile = rem * (rem + 1) / 2;       // u = 0 is tile (k+1,k+1): wave 0's
        int u = wave;                                // waves 1..7 -> u = 1.., stride 7
        Frag f0, f1;
        d4 c0, c1;
        int i0 = 0, j0 = 0, i1 = 0, j1 = 0;
        if (u < ntile) {
          tri_decode(u, k + 1, i0, j0);
          frag_load(S, tile_L(i0, k), tile_L(j0, k), lane, f0);
          c0 = tile_load(S, tile_L(i0, j0), lane);
        }
        while (u < ntile) {
          const int u1 = u + (NW - 1);
          if (u1 < ntile) {
            tri_decode(u1, k + 1, i1, j1);
            frag_load(S, tile_L(i1, k), tile_L(j1, k), lane, f1);
            c1 = tile_load(S, tile_L(i1, j1), lane);
          }
          c0 = frag_mma<true>(f0, c0);
          tile_store(S, tile_L(i0, j0), lane, c0);
          if (u1 >= ntile) break;
          const int u2 = u1 + (NW - 1);
          if (u2 < ntile) {
            tri_decode(u2, k + 1, i0, j0);
            frag_load(S, tile_L(i0, k), tile_L(j0, k), lane, f0);
            c0 = tile_load(S, tile_L(i0, j0), lane);
          }
          c1 = frag_mma<true>(f1, c1);
          tile_store(S, tile_L(i1, j1), lane, c1);
          u = u2;
        }
        // idle time of waves 1..7 while wave 0 runs the pivot chain: the parts of X = L^-1 whose inputs are
        // already final (full 128 leaf only; inputs were completed before the barrier that opened this phase)
        if (overlap_inv) {
          if ((k & 1) && wave == 1) inv_level1(S, (k - 1) >> 1, lane);             // k = 1, 3, 5: nodes 0, 1, 2
          if (k == 4 && wave <= 4) inv_level2(S, 0, (wave - 1) >> 1, (wave - 1) & 1, lane);
          if (k == 5 && wave >= 2) {                                               // 16 T tiles on waves 2..7
            for (int id = wave - 2; id < 16; id += NW - 2) inv_level3_T(S, id & 3, id >> 2, lane);
          }
        }
      }
      __syncthreads();
    }
  }
  const long long t_factored = dbg ? wall_clock64() : 0;

  // ---- off-diagonal tiles of X by recursive doubling:  X21 = -X22 (L21 X11) ----------------------------------
  if (overlap_inv) {
    // levels 1 (nodes 0..2), 2 (node 0) and the T phase of level 3 were done in the shadow of the pivot chain
    if (wave == 0) inv_level1(S, 3, lane);
    __syncthreads();
    if (wave < 4) inv_level2(S, 1, wave >> 1, wave & 1, lane);
    __syncthreads();
  } else {
    if (wave < 4) inv_level1(S, wave, lane);
    __syncthreads();
    inv_level2(S, wave >> 2, (wave >> 1) & 1, wave & 1, lane);
    __syncthreads();
    // level 3 phase 1: 16 T tiles, two per wave
#pragma unroll
    for (int h = 0; h < 2; ++h) inv_level3_T(S, 2 * (wave & 1) + h, wave >> 1, lane);
    __syncthreads();
  }
  {
    // level 3 phase 2: 16 tiles, two per wave, pairing heavy with light rows: wave -> b = wave>>1, a in (1,2) or (0,3)
    d4 rr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = wave >> 1;
      const int a = (wave & 1) ? (1 + h) : (3 * h);
      rr[h] = inv_level3_X(S, a, b, lane);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = wave >> 1;
      const int a = (wave & 1) ? (1 + h) : (3 * h);
      tile_store(S, tile_X(4 + a, b), lane, rr[h]);
    }
  }
  __syncthreads();
  const long long t_inverted = dbg ? wall_clock64() : 0;

  // ---- write L (lower triangle, valid part) and the inverse block ------------------------------------
  {
    const int jp = 2 * (tid & 63), r0 = tid >> 6;
    const bool vec = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    constexpr int NIT = NB / NW;
    const double* __restrict__ Xd = S + XD_OFF;
#pragma unroll 4
    for (int it = 0; it < NIT; ++it) {
      const int i = r0 + NW * it;
      if (!FACTORED && i < nb && jp <= i) {
        const d2 lv = *reinterpret_cast<const d2*>(&S[i * LD + jp]);
        double* dst = A + (long)i * lda + jp;
        if constexpr (WT) {
          __hip_atomic_store(dst, lv.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (jp + 1 <= i) __hip_atomic_store(dst + 1, lv.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          if (vec && jp + 1 <= i) *reinterpret_cast<d2*>(dst) = lv;
          else { dst[0] = lv.x; if (jp + 1 <= i) dst[1] = lv.y; }
        }
      }
      auto xval = [&](int j) -> double {
        if (j > i) return 0.0;
        if ((j >> 4) == (i >> 4)) return Xd[(i >> 4) * (SB * XLD) + (i & 15) * XLD + (j & 15)];
        return S[j * LD + i];
      };
      d2 xv;
      xv.x = xval(jp);
      xv.y = xval(jp + 1);
      if constexpr (WT) {
        __hip_atomic_store(&inv[i * NB + jp], xv.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&inv[i * NB + jp + 1], xv.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        *reinterpret_cast<d2*>(&inv[i * NB + jp]) = xv;
      }
    }
  }
  if (dbg && tid == 0) {
    const long long t_end = wall_clock64();
    dbg[0] = t_loaded - t_begin; dbg[1] = t_factored - t_loaded; dbg[2] = t_inverted - t_factored;
    dbg[3] = t_end - t_inverted; dbg[4] = t_end - t_begin; dbg[5] = t_begin;
  }
  if (!FACTORED && info) {
    // bad_col is wave-0 state; lane 0 of wave 0 reports (first failing pivot of the matrix wins)
    // (the leaf of column 0 resets the word: see leaf2_device.h)
    if (tid == 0) {
      const int v = bad_col >= 0 ? col0 + bad_col + 1 : 0;
      if (col0 == 0) info[0] = v;
      else if (v != 0 && info[0] == 0) info[0] = v;
    }
  }
}

}  // namespace gpk_leaf
