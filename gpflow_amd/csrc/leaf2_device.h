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
//   helper waves 2 .. 7 do everything else, two phases per diagonal tile k ("window k"):
//     alpha   L(i,k) = A(i,k) X_kk^T for the rows below the chain's two, and the last product of row block k of the inverse,
//             X(k,j) = -X_kk T(k,j): both are final results and go to global memory straight from the registers;
//     gamma   LEFT-LOOKING, each tile one multi-term product accumulated in registers (an LDS tile is read once and written once;
//             the right-looking form of the first version was bound by LDS bandwidth, 20 LDS operations per tile product):
//             the three tiles of row k+3 through column block k (what the trailing wave of tile k+2 picks up), the tiles of
//             column block k+1 below them, row block k+1 of T(i,j) = sum_{t=j}^{i-1} L(i,t) X(t,j).  All of them end with term
//             t = k; the terms t < k need nothing of window k and are accumulated BEFORE the helper waits for A_k ("early terms"),
//             so that after A_k only one product per task is left.  Tasks are dealt to the helpers, two slots each, by weight
//             (longest first) when the kernel starts.  Then the stores of the chain's tiles of column block k and the zeros right
//             of row block k of the inverse.
//   After the last diagonal tile one tile product per wave and two tiles of stores remain (round 5: three levels of recursive
//   doubling, 3.7 us, and a 2.9-us store phase).
//
//   Synchronisation (LDS words, monotonic within the kernel; every wait is bounded and a wait that expires marks the leaf as failed):
//     flag     panels published by the pivot wave                     tdone   tiles finished by the trailing wave (X_kk, L(k+1,k))
//     pdone    diagonal tiles stored by the pivot wave                lrdone  L(k+2,k) stored by the trailing wave of tile k+1
//     hdone    helpers that finished a window (6 per window)          hB      helpers that finished an alpha phase
//     rowrdy   finished tiles of the row the next trailing wave needs (3 per window)   row2   helpers that have rows 32 .. 47 in LDS
//
//   The 4 x 4 inverse of a panel is a forward substitution carried out for its four columns at once, one column per 16-lane group
//   (10 VALU instructions + 3 selects; round 5: 16 + 10 selects), and the pivot test is one compare per diagonal tile on the
//   diagonal of the finished factor (a non-positive pivot turns the rest of its tile into NaN).
#pragma once
#include "leaf_device.h"

namespace gpk_leaf2 {
using namespace gpk_leaf;

constexpr int NCHAIN = 2;             // chain waves 0, 1
constexpr int NHELP = NW - NCHAIN;    // helper waves 2 .. 7
constexpr int HT = NHELP * 64;        // helper threads
// LDS: the round-5 image (block + dense diagonal tiles of X) + the published panels (4 x (Y, lp) x 64 lanes) + sync words
constexpr int PUB_OFF = LDS_DOUBLES;
constexpr int SYNC_OFF = PUB_OFF + 4 * 128;      // 16 ints
constexpr int TAB_W = 2 * NHELP;                 // gamma tasks of one window: two slots per helper
constexpr size_t LEAF2_LDS = (size_t)(SYNC_OFF + 8) * sizeof(double);
static_assert(LEAF2_LDS <= 160 * 1024, "leaf LDS");
enum { W_FLAG = 0, W_PDONE, W_TDONE, W_LRDONE, W_HDONE, W_HB, W_ROWRDY, W_BAD0, W_BAD1, W_TIMEOUT, W_ROW2 };

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
// register tile v (v[e] of lane (c, g) = element [g+4e][c]) -> global
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

// alpha task t of window k: rows k+3 .. of column block k, then row block k of the inverse
__device__ __forceinline__ void run_alpha(double* __restrict__ S, int t, int k, int n8, int lane, const Out& o) {
  const int nB = n8 - (k + 3) > 0 ? n8 - (k + 3) : 0;
  Frag f;
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  if (t < nB) {            // L(i,k) = A(i,k) X_kk^T
    const int i = k + 3 + t;
    frag_load(S, tile_L(i, k), tile_Xd(k), lane, f);
    acc = frag_mma<false>(f, acc);
    tile_store(S, tile_L(i, k), lane, acc);
    store_L_tile(o, i, k, acc, lane);
  } else {                 // X(k,j) = -X_kk T(k,j)
    const int j = t - nB;
    frag_load(S, tile_Xd(k), tr(tile_X(k, j)), lane, f);
    acc = frag_mma<true>(f, acc);
    tile_store(S, tile_X(k, j), lane, acc);
    store_X_tile(o, k, j, acc, lane);
  }
}

// gamma tasks of window k, packed (kind << 8 | i << 4 | j; 0 = empty slot):
//   kind 1   tile (i,j) = A(i,j) - sum_{t=0}^{k} L(i,t) L(j,t)^T        row k+3: j = k+1, k+2, k+3; column block k+1: i = k+4 ..
//   kind 2   T(k+1,j) = sum_{t=j}^{k} L(k+1,t) X(t,j),  j = 0 .. k
// They are dealt to the helpers' slots at COMPILE time, for every block size: longest first, each to the helper with the least work
// so far (the two helpers that share a SIMD with a chain wave count their work 3/2).
struct TaskTable { int t[NSB + 1][NSB - 1][TAB_W]; };
constexpr TaskTable make_task_table() {
  TaskTable T{};
  for (int n8 = 1; n8 <= NSB; ++n8)
    for (int k = 0; k + 1 < n8; ++k) {
      int load2[NHELP] = {}, nslot[NHELP] = {};   // work in products, slots used
      const int nrow = (k + 3 < n8) ? 3 : 0, ncol = n8 - (k + 4) > 0 ? n8 - (k + 4) : 0;
      const int ntask = nrow + ncol + (k + 1);
      for (int t = 0; t < ntask; ++t) {   // (already in non-increasing order of weight)
        int task = 0, w = 0;
        if (t < nrow) { task = (1 << 8) | ((k + 3) << 4) | (k + 1 + t); w = k + 1; }
        else if (t < nrow + ncol) { task = (1 << 8) | ((k + 4 + t - nrow) << 4) | (k + 1); w = k + 1; }
        else { const int j = t - nrow - ncol; task = (2 << 8) | ((k + 1) << 4) | j; w = k + 1 - j; }
        int best = -1, bestc = 1 << 30;
        for (int q = 0; q < NHELP; ++q) {
          if (nslot[q] >= 2) continue;
          const int cst = (load2[q] + w) * ((q == 2 || q == 3) ? 3 : 2);
          if (cst < bestc) { bestc = cst; best = q; }
        }
        T.t[n8][k][2 * best + nslot[best]] = task;
        nslot[best] += 1;
        load2[best] += w;
      }
    }
  return T;
}
static __device__ const TaskTable g_leaf2_tasks = make_task_table();

__device__ __forceinline__ d4 add4(d4 a, d4 b) { return (d4){a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}; }
__device__ __forceinline__ d4 sub4(d4 a, d4 b) { return (d4){a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]}; }

// One multi-term task.  Both kinds accumulate  sum_t  tile(i,t) * tile(j,t)^T  over ROW-MAJOR tiles of the LDS block: for the
// inverse, X(t,j) is parked transposed in the upper triangle, i.e. AS tile (j,t); only its first term (t == j) reads the dense
// diagonal tile X(j,j) instead.  Two accumulators (kk even / odd): the four MFMAs of a term issue back to back.
struct HTask {
  int kind, i, j;      // wave-uniform
  d4 p0, p1;
};
struct HFrag { double a[4], b[4]; };
__device__ __forceinline__ void htask_load(const double* __restrict__ S, const HTask& h, int t, int lane, HFrag& f) {
  const int r = lane & 15, kq = lane >> 4;
  const double* __restrict__ pa = S + (h.i * SB + r) * LD + t * SB + kq;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) f.a[kk] = pa[4 * kk];
  if (h.kind == 2 && t == h.j) {   // X(j,j)[q][n] as B[n][q]: dense tile, transposed view
    const double* __restrict__ pb = S + XD_OFF + h.j * (SB * XLD) + kq * XLD + r;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f.b[kk] = pb[4 * kk * XLD];
  } else {
    const double* __restrict__ pb = S + (h.j * SB + r) * LD + t * SB + kq;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f.b[kk] = pb[4 * kk];
  }
}
__device__ __forceinline__ void htask_mma(HTask& h, const HFrag& f) {
  h.p0 = mfma4(f.a[0], f.b[0], h.p0);
  h.p1 = mfma4(f.a[1], f.b[1], h.p1);
  h.p0 = mfma4(f.a[2], f.b[2], h.p0);
  h.p1 = mfma4(f.a[3], f.b[3], h.p1);
}
// terms t0 .. t1 (inclusive), operands of term t+1 in flight during the MFMAs of term t
__device__ __forceinline__ void htask_terms(const double* __restrict__ S, HTask& h, int t0, int t1, int lane) {
  if (h.kind == 0 || t0 > t1) return;
  HFrag f, n;
  htask_load(S, h, t0, lane, f);
  int t = t0;
  for (; t + 2 <= t1; t += 2) {
    htask_load(S, h, t + 1, lane, n);
    htask_mma(h, f);
    htask_load(S, h, t + 2, lane, f);
    htask_mma(h, n);
  }
  if (t + 1 <= t1) {
    htask_load(S, h, t + 1, lane, n);
    htask_mma(h, f);
    htask_mma(h, n);
  } else {
    htask_mma(h, f);
  }
}
__device__ __forceinline__ void htask_open(HTask& h, int packed) {
  h.kind = packed >> 8; h.i = (packed >> 4) & 15; h.j = packed & 15;
  h.p0 = (d4){0.0, 0.0, 0.0, 0.0}; h.p1 = h.p0;
}
__device__ __forceinline__ void htask_close(double* __restrict__ S, const HTask& h, int lane) {
  if (h.kind == 1) tile_store(S, tile_L(h.i, h.j), lane, sub4(tile_load(S, tile_L(h.i, h.j), lane), add4(h.p0, h.p1)));
  else if (h.kind == 2) tile_store(S, tile_X(h.i, h.j), lane, add4(h.p0, h.p1));
}

// The leaf (NOT already factored): one workgroup of NT threads, LDS block S of LEAF2_LDS bytes.
// STAMPS (tools/leaf_probe.hip only): shader-clock time stamps.  dbg[16 + 8 k + q], written by the PIVOT wave of tile k: 0 tile
// started, 1 .. 3 panels 0 .. 2 done, 4 panel 3 done and L stored; dbg[80 + 8 k + q], by the TRAILING wave of tile k: 0 start, 1
// inputs there (waits done), 2 L(k+1,k-1) stored, 3 look-ahead pair formed, 4 last panel followed (s2 final); dbg[144 + 8 k + q],
// helper wave 2 in window k: 0 window entered, 1 early terms done, 2 A_k reached, 3 alpha done, 4 B_k reached, 5 last terms done,
// 6 stores issued.
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
  if (tid < 16) sync[tid] = (tid == W_BAD0 || tid == W_BAD1) ? 0x7fffffff : 0;
  __syncthreads();   // the only workgroup barrier

  if (wave < NCHAIN) {
    // ================================================= chain waves =================================================
    __builtin_amdgcn_s_setprio(3);
    d4 d = {0.0, 0.0, 0.0, 0.0}, a2 = d, s2 = d;
    // tiles (0,0) [wave 0] and (1,0), (1,1) [wave 1] straight from global memory (identity beyond nb)
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
            if (k == 1) word_wait(sync, W_ROW2, NHELP); else word_wait(sync, W_ROWRDY, 3 * (k - 1));
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
    const int h = wave - NCHAIN;         // 0 .. 5
    const int htid = tid - 64 * NCHAIN;  // 0 .. 383
    const int rows_used = n8 * SB;
    // load: lower triangle of A from row 32 on (identity beyond nb) into LDS -- tiles (0,0), (1,0), (1,1) go to the chain waves'
    // registers only; all loads issued up front
    {
      const bool vec = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
      constexpr int ROW0 = 2 * SB;
      constexpr int PAIRS = (NB - ROW0) * (NB / 2);  // (row, column pair)
      constexpr int NIT = (PAIRS + HT - 1) / HT;     // 16
      d2 v[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e = htid + HT * it;
        const int i = ROW0 + e / (NB / 2), jp = 2 * (e % (NB / 2));
        v[it] = (d2){0.0, 0.0};
        if (i < nb && jp <= i) {
          const double* src = A + (long)i * lda + jp;
          if (vec && jp + 1 < nb) v[it] = *reinterpret_cast<const d2*>(src);
          else { v[it].x = src[0]; if (jp + 1 < nb) v[it].y = src[1]; }
        }
      }
      // while they fly: for a ragged block the identity rows of the inverse
      if (n8 < NSB) {
        for (int e = htid; e < NB * (NB / 2); e += HT) {
          const int i = e / (NB / 2), jp = 2 * (e % (NB / 2));
          if (i >= rows_used) {
            d2 z = {(jp == i) ? 1.0 : 0.0, (jp + 1 == i) ? 1.0 : 0.0};
            *reinterpret_cast<d2*>(&inv[i * NB + jp]) = z;
          }
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e = htid + HT * it;
        const int i = ROW0 + e / (NB / 2), jp = 2 * (e % (NB / 2));
        if (jp <= i && i < rows_used) {
          d2 w = v[it];
          if (i >= nb) w.x = (jp == i) ? 1.0 : 0.0;
          if (jp + 1 > i) w.y = 0.0; else if (i >= nb) w.y = (jp + 1 == i) ? 1.0 : 0.0;
          *reinterpret_cast<d2*>(&S[i * LD + jp]) = w;
        }
        if (it == 2) word_add(sync + W_ROW2, lane);   // rows 32 .. 49: what the trailing wave of tile 1 needs
      }
    }
    word_add(sync + W_HDONE, lane);
    for (int k = 0; k + 1 < n8; ++k) {
      auto stamp = [&](int q) { if constexpr (STAMPS) { if (wave == NCHAIN && lane == 0) dbg[144 + 8 * k + q] = clock64(); } };
      stamp(0);
      // ---- early terms of this window's tasks (t < k): everything they read was final at B_{k-1}
      HTask ta, tb;
      htask_open(ta, g_leaf2_tasks.t[n8][k][2 * h]);
      htask_open(tb, g_leaf2_tasks.t[n8][k][2 * h + 1]);
      htask_terms(S, ta, ta.kind == 2 ? ta.j : 0, k - 1, lane);
      htask_terms(S, tb, tb.kind == 2 ? tb.j : 0, k - 1, lane);
      stamp(1);
      // ---- A_k: every helper is through window k-1, the chain has stored L(k,k), X_kk, L(k+1,k)
      word_wait(sync, W_HDONE, NHELP * (k + 1));
      word_wait(sync, W_PDONE, k + 1);
      word_wait(sync, W_TDONE, k + 1);
      stamp(2);
      // ---- alpha
      {
        const int nB = n8 - (k + 3) > 0 ? n8 - (k + 3) : 0;
        for (int t = h; t < nB + k; t += NHELP) run_alpha(S, t, k, n8, lane, out);
      }
      word_add(sync + W_HB, lane);
      stamp(3);
      // ---- B_k: alpha of every helper is done, the trailing wave of tile k+1 has stored L(k+2,k)
      word_wait(sync, W_HB, NHELP * (k + 1));
      if (k + 2 < n8) word_wait(sync, W_LRDONE, k + 1);
      stamp(4);
      // ---- last term (t = k), results to LDS; the row tasks (what the next trailing wave waits for) sit in slot 0
      htask_terms(S, ta, k, k, lane);
      htask_close(S, ta, lane);
      if (ta.kind == 1 && ta.i == k + 3) word_add(sync + W_ROWRDY, lane);
      htask_terms(S, tb, k, k, lane);
      htask_close(S, tb, lane);
      if (tb.kind == 1 && tb.i == k + 3) word_add(sync + W_ROWRDY, lane);
      stamp(5);
      // ---- the chain's tiles of column block k (L(k,k), L(k+1,k), L(k+2,k), X_kk) and the zeros right of row block k of the inverse
      for (int u = h; u < 4; u += NHELP) {
        if (u < 3) { if (k + u < n8) store_L_tile(out, k + u, k, tile_load(S, tile_L(k + u, k), lane), lane); }
        else store_X_tile(out, k, k, tile_load(S, tile_Xd(k), lane), lane);
      }
      const int zp = (NB - (k + 1) * SB) / 2;   // column pairs right of the diagonal tile
      for (int r = h; r < SB; r += NHELP)
        for (int q = lane; q < zp; q += 64) *reinterpret_cast<d2*>(&inv[(k * SB + r) * NB + (k + 1) * SB + 2 * q]) = (d2){0.0, 0.0};
      stamp(6);
      word_add(sync + W_HDONE, lane);
    }
  }
  // ================================== tail (all eight waves): last row block of the inverse ==================================
  {
    const int k = n8 - 1;
    word_wait(sync, W_HDONE, NHELP * n8);
    word_wait(sync, W_PDONE, n8);
    word_wait(sync, W_TDONE, n8);
    const long long t_factored = dbg ? wall_clock64() : 0;
    if (n8 < NSB) {
      const int zp = (NB - (k + 1) * SB) / 2;
      for (int r = wave; r < SB; r += NW)
        for (int q = lane; q < zp; q += 64) *reinterpret_cast<d2*>(&inv[(k * SB + r) * NB + (k + 1) * SB + 2 * q]) = (d2){0.0, 0.0};
    }
    for (int t = wave; t < k + 2; t += NW) {
      if (t < k) run_alpha(S, t, k, n8, lane, out);   // (no rows below: task t is X(k,t) = -X_kk T(k,t))
      else if (t == k) store_L_tile(out, k, k, tile_load(S, tile_L(k, k), lane), lane);
      else store_X_tile(out, k, k, tile_load(S, tile_Xd(k), lane), lane);
    }
    if (wave == 0 && lane == 0 && info) {                // first failing pivot of the matrix wins (an earlier leaf may have reported)
      const int b0 = sync[W_BAD0], b1 = sync[W_BAD1];
      const int b = b0 < b1 ? b0 : b1;
      if (info[0] == 0) {
        if (b != 0x7fffffff) info[0] = col0 + b + 1;
        else if (sync[W_TIMEOUT]) info[0] = 0x7fffffff;  // a hand-off inside the leaf timed out (gpk.h: INT_MAX)
      }
    }
    if (dbg && tid == 0) {
      const long long t_end = wall_clock64();
      dbg[0] = 0; dbg[1] = t_factored - t_begin; dbg[2] = 0;
      dbg[3] = t_end - t_factored; dbg[4] = t_end - t_begin; dbg[5] = t_begin;
    }
  }
}

}  // namespace gpk_leaf2
