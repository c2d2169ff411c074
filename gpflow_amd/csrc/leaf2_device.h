// Round-6 leaf: the 128 x 128 diagonal-block factorisation A = L L^T, X = L^-1 as a two-wave pivot pipeline plus six helper waves,
// synchronised through LDS counters -- no workgroup barrier after the first instruction.
// (Lane-level model and hazard check of the schedule: tools/leaf2_model.py; probe with in-kernel time stamps: tools/leaf_probe.hip.
//  The round 1 - 5 leaf, leaf_device.h, still inverts an existing factor's diagonal block -- FACTORED -- and is what the A/B-only
//  single-launch step kernel runs.)
//
// What was measured on the old leaf (32 us alone: load 3.4, factor 22, invert 3.7, store 2.9): wave 0 factors the 16 x 16 diagonal
// tiles, ~100 wave-uniform VALU instructions per 4-column panel (~900 cycles), and between two diagonal tiles it went through two
// workgroup barriers, two LDS round trips and two dependent tile products (L(k+1,k) = A(k+1,k) X_kk^T, then the update of tile
// (k+1,k+1)); the inverse was assembled after the factorisation; everything was written out at the very end.  Here:
//
//   chain waves 0 and 1 alternate, tile by tile, as
//     PIVOT wave    factors diagonal tile k in registers, four 4-column panels: gathers the 4 x 4 pivot block (v_readlane), factors
//                   it with wave-uniform arithmetic, forms its inverse Y as an MFMA A-operand, scales the panel (lp) and applies the
//                   rank-4 update to the tile -- and PUBLISHES (Y, lp) of each panel through LDS with a flag word.  It waits for
//                   nobody;
//     TRAILING wave holds THREE more tiles in registers: the inverse of tile k (x), the tile below it transposed in the MFMA C/D
//                   layout (a2 = A(k+1,k)) and the next diagonal tile (s2 = A(k+1,k+1)).  For every published panel it does the five
//                   MFMAs that need nothing but (Y, lp): new rows of x, panel of a2, rank-4 updates of a2 / s2 / x.  When the
//                   pivot wave publishes its last Y, two MFMAs later s2 IS the fully updated next diagonal tile, in registers, in
//                   the layout the pivot arithmetic wants: the trailing wave carries straight on as the pivot wave of tile k+1.
//     The former pivot wave becomes the trailing wave of tile k+1: it forms L(k+2,k) = A(k+2,k) X_kk^T and the look-ahead pair
//     a2 = A(k+2,k+1) - L(k+2,k) L(k+1,k)^T, s2 = A(k+2,k+2) - L(k+2,k) L(k+2,k)^T itself (12 MFMAs, results stay in registers)
//     and then catches up with the panels already published.  The critical path per diagonal tile is four panels of pivot
//     arithmetic plus ~2 MFMA latencies of hand-over -- no barrier, LDS round trip or tile product is on it.
//
//   helper waves 2 .. 11 OWN TILE ROWS of the block -- rows 7, 6, 5 two waves each (one for the row of L and the look-ahead tiles,
//   steps a, d, e below; one for the row of the inverse, step f), rows 4, 3, 2 one wave each; the heavy ones on the SIMDs without a
//   chain wave -- and keep everything of their row in registers -- its finished tiles L(I,s) as MFMA operands, the running sums
//   T(I,j) = sum_{t=j}^{I-1} L(I,t) X(t,j) of its row of the inverse, the three tiles the chain picks up.  Per diagonal tile t
//   ("window t") an owner
//     (a) accumulates  C = sum_{s<t} L(I,s) L(t,s)^T  (operands final one window earlier; nothing to wait for),
//     (d) once X_tt is published forms  L(I,t) = (A(I,t) - C) X_tt^T  in one MFMA chain, in fragment layout, without an LDS round
//         trip, and stores it (LDS for the other rows, global memory: it is final),
//     (e) adds term t to the look-ahead tiles (I,I-2), (I,I-1), (I,I); after term I-3 they go to LDS and the trailing wave of
//         tile I-1 is released (one flag per row),
//     (f) adds term t to its T(I,j), j <= t: t+1 independent MFMA chains -- the work on the inverse is spread evenly over the
//         windows (the just-in-time form of the previous version had 28 of 112 products in the last window); after term I-1 the
//         sums are parked in LDS and every helper finalises one tile X(I,j) = -X_II T(I,j) when X_II appears.
//   There is no barrier between helpers: an owner waits for the chain's flags, for "column block t of L complete" before (e), and
//   for "row block t of X complete" before (f).
//   (Versions 2 - 6 of this file, all bit-correct and measured with tools/leaf_probe.hip: barrier phases with right-looking
//    read-modify-write tile products were bound by LDS bandwidth (20 LDS operations per product) and by scalar task decoding;
//    left-looking multi-term tasks by the 160-cycle latency of a dependent v_mfma_f64_16x16x4 and by the imbalance between the
//    SIMDs; profiles/r06_leaf_probe_*.txt.)
//
//   Synchronisation (LDS words, monotonic within the kernel; every wait is bounded and a wait that expires marks the leaf as failed):
//     flag     panels published by the pivot wave            tdone   tiles finished by the trailing wave (X_kk, L(k+1,k))
//     pdone    diagonal tiles stored by the pivot wave       lrdone  L(k+2,k) stored by the trailing wave of tile k+1
//     lcol[t]  tiles of column block t stored by owners      xrowc[t]  finalised tiles of row block t of the inverse
//     tpark[t] row block t of T parked                       rowf[i]   look-ahead tiles of row i in LDS
//     row2 / loaded   helpers that have rows 32 .. 49 / all rows in LDS
//
//   The 4 x 4 inverse of a panel is a forward substitution carried out for its four columns at once, one column per 16-lane group
//   (10 VALU instructions + 3 selects; round 5: 16 + 10 selects), and the pivot test is one compare per diagonal tile on the
//   diagonal of the finished factor (a non-positive pivot turns the rest of its tile into NaN).
#pragma once
#include "leaf_device.h"

namespace gpk_leaf2 {
using namespace gpk_leaf;

#ifndef LEAF2_CHAIN_PRIO
#define LEAF2_CHAIN_PRIO 3
#endif
constexpr int NT2 = 768;              // twelve waves, three per SIMD
constexpr int NW2 = NT2 / 64;
constexpr int NCHAIN = 2;             // chain waves 0, 1
constexpr int NHELP = NW2 - NCHAIN;   // helper waves 2 .. 11
constexpr int HT = NHELP * 64;        // helper threads
// LDS: the round-5 image (block + dense diagonal tiles of X) + the published panels (4 x (Y, lp) x 64 lanes) + sync words
constexpr int PUB_OFF = LDS_DOUBLES;
constexpr int SYNC_OFF = PUB_OFF + 4 * 128;      // 48 ints
constexpr size_t LEAF2_LDS_USED = (size_t)(SYNC_OFF + 24) * sizeof(double);
// The launch asks for a WHOLE compute unit's LDS (160 KB): with the 151 KB it uses, workgroups of LDS-light kernels (the covariance
// builder beside the first leaves of an SVGP step) were placed on the leaf's CU and took issue slots from its chain waves -- the
// first two leaves of a step ran 50 us instead of 25 (profiles/r06_step_timeline.txt).
constexpr size_t LEAF2_LDS = 160 * 1024;
static_assert(LEAF2_LDS_USED <= LEAF2_LDS, "leaf LDS");
enum { W_FLAG = 0, W_PDONE, W_TDONE, W_LRDONE, W_ROW2, W_LOADED, W_BAD0, W_BAD1, W_TIMEOUT, W_LCOL = 16, W_XROWC = 24, W_TPARK = 32,
       W_ROWF = 40, W_NWORDS = 48 };

__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
// (LDS operations of one wave execute in order: data stores issued before a flag store / counter add are visible before it)
__device__ __forceinline__ void word_set(int* w, int v) {
  compiler_fence();
  __atomic_store_n(w, v, __ATOMIC_RELAXED);
}
__device__ __forceinline__ void word_add(int* w, int lane) {
  compiler_fence();
  if (lane == 0) __atomic_fetch_add(w, 1, __ATOMIC_RELAXED);
}
constexpr int SPIN_MAX = 1 << 17;
__device__ __forceinline__ void word_wait(int* sync, int idx, int want) {
  int spin = 0;
  while (__builtin_amdgcn_readfirstlane(__atomic_load_n(sync + idx, __ATOMIC_RELAXED)) < want) {
    if (++spin > SPIN_MAX) { __atomic_store_n(sync + W_TIMEOUT, 1, __ATOMIC_RELAXED); break; }
  }
  compiler_fence();
}
// A wave's SNAPSHOT of all sync words (lane l holds word l): one LDS round trip (~250 cycles here) serves every check until a word
// is found too small -- the words only grow, so a stale snapshot errs on the safe side.
struct Snap {
  int v;
  __device__ __forceinline__ void refresh(int* sync, int lane) {
    compiler_fence();
    v = __atomic_load_n(sync + (lane < W_NWORDS ? lane : 0), __ATOMIC_RELAXED);
    compiler_fence();
  }
  __device__ __forceinline__ void need(int* sync, int lane, int idx, int want) {
    int spin = 0;
    while (__builtin_amdgcn_readlane(v, idx) < want) {
      refresh(sync, lane);
      if (++spin > SPIN_MAX) { __atomic_store_n(sync + W_TIMEOUT, 1, __ATOMIC_RELAXED); break; }
    }
  }
};

// ---- pivot wave: one 4-column panel of the diagonal tile held in d (d[e] of lane (c, g) = S[g+4e][c], symmetric) --------------
// publishes yop = inv(L4) as an A-operand (lanes m = c < 4, k = g) and lp (register 0 of lane (n, g) = L[n][4P+g])
template <int P>
__device__ __forceinline__ void panel_pivot(d4& d, int c, int g, int lane, const double (&ind)[4], double* __restrict__ pub,
                                            int* __restrict__ sync, int seq0) {
  auto pick = [&](int a, int b) -> double { return readlane_d(d[P], 4 * P + b + 16 * a); };
  const double s00 = pick(0, 0), s10 = pick(1, 0), s20 = pick(2, 0), s30 = pick(3, 0);
  const double s11 = pick(1, 1), s21 = pick(2, 1), s31 = pick(3, 1);
  const double s22 = pick(2, 2), s32 = pick(3, 2), s33 = pick(3, 3);
  const double r0 = rsqrt_nr(s00);
  const double l10 = s10 * r0, l20 = s20 * r0, l30 = s30 * r0;
  const double r1 = rsqrt_nr(fma(-l10, l10, s11));
  const double l21 = fma(-l20, l10, s21) * r1;
  const double l31 = fma(-l30, l10, s31) * r1;
  const double r2 = rsqrt_nr(fma(-l21, l21, fma(-l20, l20, s22)));
  const double l32 = fma(-l31, l21, fma(-l30, l20, s32)) * r2;
  const double r3 = rsqrt_nr(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, s33))));
  // L4 y = e_g by forward substitution, column g of Y in lane group g (ind[k] = 1.0 in the lanes g == k, c < 4, else 0.0)
  const double y0 = r0 * ind[0];
  const double y1 = r1 * fma(-l10, y0, ind[1]);
  const double y2 = r2 * fma(-l21, y1, fma(-l20, y0, ind[2]));
  const double y3 = r3 * fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, ind[3])));
  double yop = (c == 1) ? y1 : y0;
  yop = (c == 2) ? y2 : yop;
  yop = (c == 3) ? y3 : yop;
  pub[P * 128 + lane] = yop;
  word_set(sync + W_FLAG, seq0 + 2 * P + 1);
  const d4 zero = {0.0, 0.0, 0.0, 0.0};
  const d4 t = mfma4(yop, d[P], zero);
  const double lp = (c >= 4 * P + g) ? t[0] : 0.0;   // rows above the panel and the upper part of the 4 x 4 block
  if constexpr (P < 3) {
    pub[P * 128 + 64 + lane] = lp;
    word_set(sync + W_FLAG, seq0 + 2 * P + 2);
    d = mfma4(-lp, lp, d);
  }
  d[P] = lp;
}

// ---- trailing wave: the same panel on the tiles it carries (x: inverse of tile k; a2[e] of lane (c, g) = A(k+1,k)[c][4e+g];
// s2 = A(k+1,k+1) in d's layout).  One LDS round trip per poll: flag word and both published vectors are read together (the
// pivot wave stores a vector BEFORE the flag value that announces it, and LDS serves a wave's operations in order).
template <int P, bool AHEAD>
__device__ __forceinline__ void panel_trail(d4& x, d4& a2, d4& s2, int lane, const double* __restrict__ pub, int* __restrict__ sync,
                                            int seq0) {
  const d4 zero = {0.0, 0.0, 0.0, 0.0};
  int f = 0, spin = 0;
  double yop = 0.0, lp = 0.0;
  for (;;) {
    f = __atomic_load_n(sync + W_FLAG, __ATOMIC_RELAXED);
    compiler_fence();
    yop = pub[P * 128 + lane];
    lp = pub[P * 128 + 64 + lane];
    compiler_fence();
    f = __builtin_amdgcn_readfirstlane(f);
    if (f >= seq0 + 2 * P + 1) break;
    if (++spin > SPIN_MAX) { __atomic_store_n(sync + W_TIMEOUT, 1, __ATOMIC_RELAXED); break; }
  }
  double lp2 = 0.0;
  if constexpr (AHEAD) {
    const d4 t2 = mfma4(yop, a2[P], zero);
    lp2 = t2[0];                                  // lane (n, g): L(k+1,k)[n][4P+g]
  }
  const d4 u = mfma4(yop, x[P], zero);
  const double xp = u[0];                         // lane (n, g): X[4P+g][n]
  if constexpr (AHEAD) s2 = mfma4(-lp2, lp2, s2);
  if constexpr (P < 3) {
    spin = 0;
    while (f < seq0 + 2 * P + 2) {
      f = __atomic_load_n(sync + W_FLAG, __ATOMIC_RELAXED);
      compiler_fence();
      lp = pub[P * 128 + 64 + lane];
      compiler_fence();
      f = __builtin_amdgcn_readfirstlane(f);
      if (++spin > SPIN_MAX) { __atomic_store_n(sync + W_TIMEOUT, 1, __ATOMIC_RELAXED); break; }
    }
    if constexpr (AHEAD) a2 = mfma4(-lp, lp2, a2);  // A2[n][j] -= sum_q L2[n][4P+q] L[j][4P+q]
    x = mfma4(-lp, xp, x);                          // Xtmp[m][:] -= L[m][4P+q] X[4P+q][:]
  }
  x[P] = xp;
  if constexpr (AHEAD) a2[P] = lp2;
}

// ---- helpers --------------------------------------------------------------------------------------------------------------------
struct Out {
  double* __restrict__ A; long lda; int nb;
  double* __restrict__ inv;
};
// register tile v in the MFMA C/D layout (v[e] of lane (c, g) = element [g+4e][c]) -> global
__device__ __forceinline__ void store_L_tile(const Out& o, int ti, int tj, d4 v, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = ti * SB + g + 4 * e, col = tj * SB + c;
    if (row < o.nb && col <= row) o.A[(long)row * o.lda + col] = v[e];
  }
}
__device__ __forceinline__ void store_X_tile(const Out& o, int ti, int tj, d4 v, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) o.inv[(ti * SB + g + 4 * e) * NB + tj * SB + c] = v[e];
}
__device__ __forceinline__ d4 add4(d4 a, d4 b) { return (d4){a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}; }
__device__ __forceinline__ d4 sub4(d4 a, d4 b) { return (d4){a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]}; }

// FRAGMENT layout of a 16 x 16 tile M: register e of lane (c, g) = M[c][4e+g] -- an MFMA A- or B-operand (kk = e) as it stands
__device__ __forceinline__ d4 frag_rm(const double* __restrict__ S, int ti, int tj, int lane) {   // row-major tile (ti, tj) of the block
  const double* __restrict__ p = S + (ti * SB + (lane & 15)) * LD + tj * SB + (lane >> 4);
  return (d4){p[0], p[4], p[8], p[12]};
}
__device__ __forceinline__ d4 frag_xd(const double* __restrict__ S, int k, int lane) {             // X_kk
  const double* __restrict__ p = S + XD_OFF + k * (SB * XLD) + (lane & 15) * XLD + (lane >> 4);
  return (d4){p[0], p[4], p[8], p[12]};
}
__device__ __forceinline__ d4 frag_xdt(const double* __restrict__ S, int k, int lane) {            // X_kk^T
  const double* __restrict__ p = S + XD_OFF + k * (SB * XLD) + (lane >> 4) * XLD + (lane & 15);
  return (d4){p[0], p[4 * XLD], p[8 * XLD], p[12 * XLD]};
}
__device__ __forceinline__ void store_frag_rm(double* __restrict__ S, int ti, int tj, d4 v, int lane) {
  double* __restrict__ p = S + (ti * SB + (lane & 15)) * LD + tj * SB + (lane >> 4);
  p[0] = v[0]; p[4] = v[1]; p[8] = v[2]; p[12] = v[3];
}
// D[m][n] (+)= sum_q A[m][q] B[n][q] with both operands in fragment layout; kk even / odd in two accumulators (half the chain)
__device__ __forceinline__ void mma2(d4 a, d4 b, d4& p0, d4& p1) {
  p0 = mfma4(a[0], b[0], p0);
  p1 = mfma4(a[1], b[1], p1);
  p0 = mfma4(a[2], b[2], p0);
  p1 = mfma4(a[3], b[3], p1);
}
__device__ __forceinline__ d4 mma1(d4 a, d4 b, d4 acc) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) acc = mfma4(a[kk], b[kk], acc);
  return acc;
}

// duty of every helper (and, for the last row block, of the chain waves): X(t,j) = -X_tt T(t,j) for its share of row block t
__device__ __forceinline__ void finalize_tile(double* __restrict__ S, int t, int j, int lane, const Out& o, int* __restrict__ sync) {
  const d4 xd = frag_xd(S, t, lane);
  const d4 tt = frag_rm(S, j, t, lane);   // T(t,j) is parked transposed, i.e. as tile (j, t): Bnt[n][q] = T[q][n]
  d4 p0 = {0.0, 0.0, 0.0, 0.0}, p1 = p0;
  mma2(-xd, tt, p0, p1);
  const d4 x = add4(p0, p1);
  tile_store(S, tile_X(t, j), lane, x);
  store_X_tile(o, t, j, x, lane);
  word_add(sync + W_XROWC + t, lane);
}

// The owner of tile row I (compile time).  `mine` = the row exists (I < n8); the duties are done either way.
// MODE 0: the whole row; 1: its tiles of L and the look-ahead tiles (a, d, e); 2: its row of the inverse (f); 3: duties only.
// DUTY 1: the chain's tiles of column block t to global memory, and row 1 of the inverse (it has no owner).
// DUTY 2: the zeros right of row block t of the inverse.
template <int I, int MODE, int DUTY, bool STAMPS>
__device__ __forceinline__ void owner_row(double* __restrict__ S, int h, int lane, int n8, const Out& out, int* __restrict__ sync,
                                          long long* __restrict__ dbg) {
  constexpr int NL = I >= 3 ? I - 2 : 1;   // own tiles L(I,s), s = 0 .. I-3
  constexpr bool DO_L = MODE == 0 || MODE == 1, DO_X = MODE == 0 || MODE == 2;
  const bool mine = I < n8;
  const d4 zero = {0.0, 0.0, 0.0, 0.0};
  d4 Lown[NL], T[I], R[3];
#pragma unroll
  for (int s = 0; s < NL; ++s) Lown[s] = zero;
#pragma unroll
  for (int j = 0; j < I; ++j) T[j] = zero;
  R[0] = zero; R[1] = zero; R[2] = zero;
  Snap sn;
  sn.refresh(sync, lane);
  for (int t = 0; t < n8; ++t) {
    auto stamp = [&](int q) {
      if constexpr (STAMPS) { if ((h == 0 || h == 1) && lane == 0) dbg[144 + 64 * h + 8 * t + q] = clock64(); }
    };
    stamp(0);
    // ---- (a) C = sum_{s<t} L(I,s) L(t,s)^T, transposed (so that it comes out in fragment layout): D[m][n] = sum L(t,s)[m][q] L(I,s)[n][q]
    d4 c0 = zero, c1 = zero;
    const bool col = DO_L && mine && t <= I - 3;
    if (col && t >= 1) {
      sn.need(sync, lane, W_TDONE, t);                       // L(t,t-1)
      if (t >= 2) sn.need(sync, lane, W_LRDONE, t - 1);      // L(t,t-2)
      if (t >= 3) sn.need(sync, lane, W_LCOL + t - 3, n8 - t);   // L(t,t-3) (column block t-3 complete: n8 - (t-3) - 3 tiles)
#pragma unroll
      for (int s = 0; s < NL; ++s)
        if (s < t) mma2(frag_rm(S, t, s, lane), Lown[s], c0, c1);
    }
    stamp(1);
    // ---- (b) A_t: X_tt, L(t+1,t) and L(t,t) are in LDS
    sn.need(sync, lane, W_TDONE, t + 1);
    sn.need(sync, lane, W_PDONE, t + 1);
    stamp(2);
    // ---- (d) L(I,t) = (A(I,t) - C) X_tt^T:  D[m][n] = sum_q X[m][q] Cc[n][q], again in fragment layout
    d4 lr = zero;
    if (col) {
      const d4 cc = sub4(frag_rm(S, I, t, lane), add4(c0, c1));
      d4 p0 = zero, p1 = zero;
      mma2(frag_xd(S, t, lane), cc, p0, p1);
      lr = add4(p0, p1);
      store_frag_rm(S, I, t, lr, lane);
      word_add(sync + W_LCOL + t, lane);
      {  // final: to global memory (register e of lane (c, g) = L[c][4e+g])
        const int row = I * SB + (lane & 15);
        if (row < out.nb) {
          double* __restrict__ dst = out.A + (long)row * out.lda + t * SB + (lane >> 4);
          dst[0] = lr[0]; dst[4] = lr[1]; dst[8] = lr[2]; dst[12] = lr[3];
        }
      }
#pragma unroll
      for (int s = 0; s < NL; ++s)
        if (s == t) Lown[s] = lr;
    }
    stamp(3);
    // ---- (e) look-ahead tiles of row I: term t of  sum_{s <= I-3} L(I,s) L(j',s)^T,  j' = I-2, I-1, I
    if constexpr (I >= 3) {
      if (col) {
        sn.need(sync, lane, W_LCOL + t, n8 - t - 3);         // rows I-2, I-1 of column block t from their owners ...
        if (t + 2 < n8) sn.need(sync, lane, W_LRDONE, t + 1);   // ... or from the chain: L(t+2,t); L(t+1,t) came with A_t
        const d4 b0 = frag_rm(S, I - 2, t, lane), b1 = frag_rm(S, I - 1, t, lane);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          R[0] = mfma4(lr[kk], b0[kk], R[0]);
          R[1] = mfma4(lr[kk], b1[kk], R[1]);
          R[2] = mfma4(lr[kk], lr[kk], R[2]);
        }
        if (t == I - 3) {
#pragma unroll
          for (int q = 0; q < 3; ++q) tile_store(S, tile_L(I, I - 2 + q), lane, sub4(tile_load(S, tile_L(I, I - 2 + q), lane), R[q]));
          word_set(sync + W_ROWF + I, 1);
        }
      }
    }
    stamp(4);
    // ---- duties of window t
    if (t >= 1) {   // finalise my share of row block t of the inverse (all eight waves share the last one)
      const bool last = t == n8 - 1;
      sn.need(sync, lane, W_TPARK + t, 1);
      for (int j = h; j < t; j += (last ? NW2 : NHELP)) finalize_tile(S, t, j, lane, out, sync);
    }
    if constexpr (DUTY == 1) {
      store_L_tile(out, t, t, tile_load(S, tile_L(t, t), lane), lane);
      store_X_tile(out, t, t, tile_load(S, tile_Xd(t), lane), lane);
      if (t + 1 < n8) store_L_tile(out, t + 1, t, tile_load(S, tile_L(t + 1, t), lane), lane);
      if (t >= 1 && t + 1 < n8) store_L_tile(out, t + 1, t - 1, tile_load(S, tile_L(t + 1, t - 1), lane), lane);   // L(t+1,t-1): stored before A_t
      if (t == 0 && n8 > 1) {   // row 1 of the inverse: T(1,0) = L(1,0) X_00
        d4 p0 = zero, p1 = zero;
        mma2(frag_rm(S, 1, 0, lane), frag_xdt(S, 0, lane), p0, p1);
        tile_store(S, tile_X(1, 0), lane, add4(p0, p1));
        word_set(sync + W_TPARK + 1, 1);
      }
    }
    if constexpr (DUTY == 2) {
      const int zp = (NB - (t + 1) * SB) / 2;   // column pairs right of the diagonal tile
      for (int r = 0; r < SB; ++r)
        for (int q = lane; q < zp; q += 64) *reinterpret_cast<d2*>(&out.inv[(t * SB + r) * NB + (t + 1) * SB + 2 * q]) = (d2){0.0, 0.0};
    }
    stamp(5);
    // ---- (f) term t of T(I,j), j <= t:  D[m][n] += sum_q L(I,t)[m][q] X(t,j)[q][n]   (X(t,j) parked transposed = tile (j,t))
    if (DO_X && mine && t < I) {
      d4 aop = lr;
      if constexpr (MODE == 2) {   // the row of L is another wave's: column block t complete, then from LDS
        if (t <= I - 3) { sn.need(sync, lane, W_LCOL + t, n8 - t - 3); aop = frag_rm(S, I, t, lane); }
      }
      if (t == I - 2) { sn.need(sync, lane, W_LRDONE, I - 1); aop = frag_rm(S, I, t, lane); }   // L(I,I-2) from the trailing wave
      if (t == I - 1) aop = frag_rm(S, I, t, lane);                                          // L(I,I-1): came with A_t
      if (t >= 1) sn.need(sync, lane, W_XROWC + t, t);
#pragma unroll
      for (int j0 = 0; j0 < I; j0 += 2) {
        if (j0 <= t) {   // pairs of tiles: two independent chains per block; the second operand is zeroed beyond the diagonal
          const d4 b0 = (j0 == t) ? frag_xdt(S, t, lane) : frag_rm(S, j0, t, lane);
          d4 b1 = zero;
          if (j0 + 1 < I) {
            if (j0 + 1 == t) b1 = frag_xdt(S, t, lane);
            else if (j0 + 1 < t) b1 = frag_rm(S, j0 + 1, t, lane);
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            T[j0] = mfma4(aop[kk], b0[kk], T[j0]);
            if (j0 + 1 < I) T[j0 + 1] = mfma4(aop[kk], b1[kk], T[j0 + 1]);
          }
        }
      }
      if (t == I - 1) {
#pragma unroll
        for (int j = 0; j < I; ++j) tile_store(S, tile_X(I, j), lane, T[j]);
        word_set(sync + W_TPARK + I, 1);
      }
    }
    stamp(6);
  }
}

// The leaf (NOT already factored): one workgroup of NT2 threads, LDS block S of LEAF2_LDS bytes.
// STAMPS (tools/leaf_probe.hip only): shader-clock time stamps.  dbg[16 + 8 k + q], written by the PIVOT wave of tile k: 0 tile
// started, 1 .. 3 panels 0 .. 2 done, 4 panel 3 done and L stored; dbg[80 + 8 k + q], by the TRAILING wave of tile k: 0 start, 1
// inputs there (waits done), 2 L(k+1,k-1) stored, 3 look-ahead pair formed, 4 last panel followed (s2 final); dbg[144 + 8 t + q], the
// owner of the inverse's row 7 in window t: 0 entered, 1 (a) done, 2 A_t reached, 3 (d) done, 4 (e) done, 5 duties done, 6 (f)
// done; dbg[208 + 8 t + q]: the same for the owner of row 7 of L; dbg[8 .. 10]: helper 0 loads issued / rows in LDS / all loaded.
template <bool STAMPS = false>
__device__ __forceinline__ void leaf2_body(double* __restrict__ S, double* __restrict__ A, long lda, int nb,
                                           double* __restrict__ inv, int* __restrict__ info, int col0,
                                           long long* __restrict__ dbg) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n8 = (nb + SB - 1) / SB;   // diagonal tiles that hold part of the matrix
  const int c = lane & 15, g = lane >> 4;
  const long long t_begin = dbg ? wall_clock64() : 0;
  const Out out{A, lda, nb, inv};
  double* __restrict__ pub = S + PUB_OFF;
  int* __restrict__ sync = reinterpret_cast<int*>(S + SYNC_OFF);
  if constexpr (STAMPS) { if (tid == 0) dbg[11] = clock64(); }
  // ---- every wave issues its global loads FIRST (~1 us of latency: everything up to the first use overlaps with it)
  constexpr int ROW0 = 2 * SB, NIT = (NB - ROW0 + NHELP - 1) / NHELP;   // helpers: 10 rows per wave, one row (1 KB) per iteration
  const int h = wave - NCHAIN;         // helper index 0 .. 9
  const int htid = tid - 64 * NCHAIN;
  const int rows_used = n8 * SB;
  const int jp = 2 * lane;
  d4 d = {0.0, 0.0, 0.0, 0.0}, a2 = d, s2 = d;
  d2 v[NIT];
  if (wave < NCHAIN) {
    // tiles (0,0) [wave 0] and (1,0), (1,1) [wave 1] straight from global memory into registers (identity beyond nb)
    auto gsym = [&](int t0, int e) -> double {   // element [g+4e][c] of the symmetric diagonal tile t0, from its lower half
      const int r = g + 4 * e;
      const int hi = t0 * SB + (r > c ? r : c), lo = t0 * SB + (r > c ? c : r);
      return (hi < nb) ? A[(long)hi * lda + lo] : (hi == lo ? 1.0 : 0.0);
    };
    if (wave == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = gsym(0, e);
    } else if (n8 > 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a2[e] = (SB + c < nb) ? A[(long)(SB + c) * lda + 4 * e + g] : 0.0;
        s2[e] = gsym(1, e);
      }
    }
  } else {
    // lower triangle of A from row 32 on (tiles (0,0), (1,0), (1,1) go to the chain waves' registers only)
    const bool vec = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    if (vec && nb == NB) {   // the usual case: no per-element edge tests, pointer increments
      const double* src = A + (long)(ROW0 + h) * lda + jp;
      const long step = (long)NHELP * lda;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = ROW0 + h + NHELP * it;
        v[it] = (d2){0.0, 0.0};
        if (i < NB && jp <= i) v[it] = *reinterpret_cast<const d2*>(src);
        src += step;
      }
    } else {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = ROW0 + h + NHELP * it;
        v[it] = (d2){0.0, 0.0};
        if (i < nb && jp <= i) {
          const double* src = A + (long)i * lda + jp;
          if (vec && jp + 1 < nb) v[it] = *reinterpret_cast<const d2*>(src);
          else { v[it].x = src[0]; if (jp + 1 < nb) v[it].y = src[1]; }
        }
      }
    }
    if constexpr (STAMPS) { if (h == 0 && lane == 0) dbg[8] = clock64(); }
  }
  if (tid < W_NWORDS) sync[tid] = (tid == W_BAD0 || tid == W_BAD1) ? 0x7fffffff : 0;
  __syncthreads();   // the only workgroup barrier
  if constexpr (STAMPS) { if (lane == 0 && wave >= NCHAIN) dbg[288 + wave - NCHAIN] = clock64(); }

  if (wave < NCHAIN) {
    // ================================================= chain waves =================================================
    __builtin_amdgcn_s_setprio(LEAF2_CHAIN_PRIO);
    double ind[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ind[q] = (g == q && c < 4) ? 1.0 : 0.0;
    int bad_col = -1;
    for (int k = 0; k < n8; ++k) {
      const int kb = k * SB, seq0 = 8 * k;
      const bool ahead = k + 1 < n8;
      if ((k & 1) == wave) {
        // ------------------------------------------------ pivot wave of tile k ------------------------------------------------
        auto stamp = [&](int q) { if constexpr (STAMPS) { if (lane == 0) dbg[16 + 8 * k + q] = clock64(); } };
        stamp(0);
        panel_pivot<0>(d, c, g, lane, ind, pub, sync, seq0);
        stamp(1);
        panel_pivot<1>(d, c, g, lane, ind, pub, sync, seq0);
        stamp(2);
        panel_pivot<2>(d, c, g, lane, ind, pub, sync, seq0);
        stamp(3);
        panel_pivot<3>(d, c, g, lane, ind, pub, sync, seq0);
        double diag = 1.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = 4 * e + g;
          if (c >= j) S[(kb + c) * LD + kb + j] = d[e];       // L[c][j]
          diag = (j == c) ? d[e] : diag;
        }
        // pivot test on the finished tile: a non-positive (or NaN) pivot leaves NaN on the diagonal from its column on
        if (bad_col < 0) {
          const unsigned long long badm = __ballot(!(diag > 0.0));
          if (badm) {
            int first = SB;
            for (int q = SB - 1; q >= 0; --q) if ((badm >> (q + 16 * (q & 3))) & 1ull) first = q;
            bad_col = kb + first;
            if (bad_col < nb && lane == 0) sync[W_BAD0 + wave] = bad_col;   // (before pdone: the reporter reads it after pdone == n8)
          }
        }
        word_set(sync + W_PDONE, k + 1);
        stamp(4);
      } else {
        // ----------------------------------------------- trailing wave of tile k -----------------------------------------------
        auto stamp = [&](int q) { if constexpr (STAMPS) { if (lane == 0) dbg[80 + 8 * k + q] = clock64(); } };
        stamp(0);
        if (k > 0) {
          // X_{k-1,k-1} and L(k,k-1) from the other chain wave; row k+1 of the block up to date through column block k-2
          word_wait(sync, W_TDONE, k);
          if (ahead) {
            if (k == 1) word_wait(sync, W_ROW2, NHELP); else word_wait(sync, W_ROWF + k + 1, 1);
          }
        }
        stamp(1);
        if (k > 0 && ahead) {
          // all operands up front
          const double* __restrict__ Xd = S + XD_OFF + (k - 1) * (SB * XLD);
          double fa[4], fb[4], fl[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = 4 * e + g;
            fa[e] = Xd[c * XLD + j];                                                             // X[c][4e+g]
            fb[e] = S[(kb + SB + c) * LD + (kb - SB) + j];                                       // A(k+1,k-1)[c][4e+g]
            fl[e] = S[(kb + c) * LD + (kb - SB) + j];                                            // L(k,k-1)[c][4e+g]
            a2[e] = S[(kb + SB + c) * LD + kb + j];                                              // A(k+1,k)[c][4e+g]
            s2[e] = (j >= c) ? S[(kb + SB + j) * LD + kb + SB + c] : S[(kb + SB + c) * LD + kb + SB + j];   // symmetric, lower half
          }
          // L(k+1,k-1) = A(k+1,k-1) X^T in FRAGMENT layout (register e of lane (n, g) = L[n][4e+g]): D[m][n] = sum_q X[m][q] A[n][q]
          d4 lr = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) lr = mfma4(fa[kk], fb[kk], lr);
#pragma unroll
          for (int e = 0; e < 4; ++e) S[(kb + SB + c) * LD + (kb - SB) + 4 * e + g] = lr[e];
          word_set(sync + W_LRDONE, k);
          stamp(2);
          // look-ahead pair of tile k:  a2 = A(k+1,k) - L(k+1,k-1) L(k,k-1)^T (transposed: D[j][n]),  s2 = A(k+1,k+1) - L L^T
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            a2 = mfma4(-fl[kk], lr[kk], a2);
            s2 = mfma4(-lr[kk], lr[kk], s2);
          }
        }
        stamp(3);
        d4 x;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (4 * e + g == c) ? 1.0 : 0.0;
        if (ahead) {
          panel_trail<0, true>(x, a2, s2, lane, pub, sync, seq0);
          panel_trail<1, true>(x, a2, s2, lane, pub, sync, seq0);
          panel_trail<2, true>(x, a2, s2, lane, pub, sync, seq0);
          panel_trail<3, true>(x, a2, s2, lane, pub, sync, seq0);
        } else {
          panel_trail<0, false>(x, a2, s2, lane, pub, sync, seq0);
          panel_trail<1, false>(x, a2, s2, lane, pub, sync, seq0);
          panel_trail<2, false>(x, a2, s2, lane, pub, sync, seq0);
          panel_trail<3, false>(x, a2, s2, lane, pub, sync, seq0);
        }
        stamp(4);
        // X_kk and L(k+1,k) to LDS for the helpers; then straight on as the pivot wave of tile k+1
        double* __restrict__ Xdk = S + XD_OFF + k * (SB * XLD);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = 4 * e + g;
          Xdk[j * XLD + c] = x[e];                              // X[j][c]
          if (ahead) S[(kb + SB + c) * LD + kb + j] = a2[e];    // L(k+1,k)[c][j]
        }
        word_set(sync + W_TDONE, k + 1);
        d = s2;
      }
    }
    __builtin_amdgcn_s_setprio(0);
  } else {
    // ===================================================== helpers =====================================================
    // for a ragged block the identity rows of the inverse; then the loaded rows into LDS (identity beyond nb)
    if (n8 < NSB) {
      for (int e = htid; e < NB * (NB / 2); e += HT) {
        const int i = e / (NB / 2), jq = 2 * (e % (NB / 2));
        if (i >= rows_used) {
          d2 z = {(jq == i) ? 1.0 : 0.0, (jq + 1 == i) ? 1.0 : 0.0};
          *reinterpret_cast<d2*>(&inv[i * NB + jq]) = z;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = ROW0 + h + NHELP * it;
      if (jp <= i && i < rows_used && i < NB) {
        d2 w = v[it];
        if (i >= nb) w.x = (jp == i) ? 1.0 : 0.0;
        if (jp + 1 > i) w.y = 0.0; else if (i >= nb) w.y = (jp + 1 == i) ? 1.0 : 0.0;
        *reinterpret_cast<d2*>(&S[i * LD + jp]) = w;
      }
      if (it == 1) word_add(sync + W_ROW2, lane);   // rows 32 .. 51: what the trailing wave of tile 1 needs
    }
    if constexpr (STAMPS) { if (lane == 0) { dbg[272 + h] = clock64(); if (h == 0) dbg[9] = clock64(); } }
    word_add(sync + W_LOADED, lane);
    word_wait(sync, W_LOADED, NHELP);
    if constexpr (STAMPS) { if (h == 0 && lane == 0) dbg[10] = clock64(); }
    // Roles (weights in tile products).  With 768 threads the waves go to SIMDs 3, 0, 2, 1, 3, 0, ... (tools/leaf_probe.hip prints
    // the HW_ID of each): the chain waves sit on SIMDs 3 and 0 and keep their fp64 VALU busy, so the helpers there (h 2, 6 and
    // 3, 7) get the work nobody waits for -- two rows of the inverse, the stores -- and everything the chain or another owner
    // waits for (rows of L, look-ahead tiles) runs on SIMDs 2 and 1:
    //   SIMD 2: h 0 inverse row 7 (33)   h 4 L row 6 (22)   h 8 row 3 (13)        SIMD 3: h 2 inverse row 5 (20)   h 6 zero fill
    //   SIMD 1: h 1 L row 7 (30)         h 5 row 4 (24)     h 9 L row 5 (15)      SIMD 0: h 3 inverse row 6 (26)   h 7 row 2 + chain stores
    switch (h) {
      case 0: owner_row<7, 2, 0, STAMPS>(S, h, lane, n8, out, sync, dbg); break;
      case 1: owner_row<7, 1, 0, STAMPS>(S, h, lane, n8, out, sync, dbg); break;
      case 2: owner_row<5, 2, 0, STAMPS>(S, h, lane, n8, out, sync, dbg); break;
      case 3: owner_row<6, 2, 0, STAMPS>(S, h, lane, n8, out, sync, dbg); break;
      case 4: owner_row<6, 1, 0, STAMPS>(S, h, lane, n8, out, sync, dbg); break;
      case 5: owner_row<4, 0, 0, STAMPS>(S, h, lane, n8, out, sync, dbg); break;
      case 6: owner_row<2, 3, 2, STAMPS>(S, h, lane, n8, out, sync, dbg); break;
      case 7: owner_row<2, 0, 1, STAMPS>(S, h, lane, n8, out, sync, dbg); break;
      case 8: owner_row<3, 0, 0, STAMPS>(S, h, lane, n8, out, sync, dbg); break;
      default: owner_row<5, 1, 0, STAMPS>(S, h, lane, n8, out, sync, dbg); break;
    }
  }
  // ================================== tail ==================================
  if (wave < NCHAIN) {
    const int t = n8 - 1;
    word_wait(sync, W_PDONE, n8);
    word_wait(sync, W_TDONE, n8);
    if (t >= 1) {   // the chain waves' share of the last row block of the inverse
      word_wait(sync, W_TPARK + t, 1);
      for (int j = NHELP + wave; j < t; j += NW2) finalize_tile(S, t, j, lane, out, sync);
    }
    if (wave == 0 && lane == 0 && info) {                // first failing pivot of the matrix wins (an earlier leaf may have reported)
      const int b0 = sync[W_BAD0], b1 = sync[W_BAD1];
      const int b = b0 < b1 ? b0 : b1;
      int v = 0;
      if (b != 0x7fffffff) v = col0 + b + 1;
      else if (sync[W_TIMEOUT]) v = 0x7fffffff;          // a hand-off inside the leaf timed out (gpk.h: INT_MAX)
      // The leaf of column 0 is the first writer of a factorisation's status word and RESETS it: no 4-byte memset packet (a fill kernel
      // plus the event the panel stream then waits for: ~20 us in the step timeline) ahead of the chain.  (Bounded waits of other streams
      // raise the word to INT_MAX after 0.5 s at the earliest.)
      if (col0 == 0) info[0] = v;
      else if (v != 0 && info[0] == 0) info[0] = v;
    }
  }
  if (dbg && tid == 0) {
    const long long t_end = wall_clock64();
    dbg[0] = 0; dbg[1] = 0; dbg[2] = 0; dbg[3] = 0; dbg[4] = t_end - t_begin; dbg[5] = t_begin;
  }
}

}  // namespace gpk_leaf2
