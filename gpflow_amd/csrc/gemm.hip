// fp64 MFMA GEMM for gfx950:  C = alpha * A * B^T + beta * C   (A [m,k], B [n,k], row-major).
//
// Every matmul-shaped piece of the dense-GP path is this one kernel:
//   * Cholesky panel solve  X = A21 * inv(L11)^T          (in place, b_tri = 2)
//   * Cholesky trailing / look-ahead updates  C -= P P^T   (c_lower, alpha=-1, beta=1)
//   * triangular solves against a cached factor            (gpk_trsm)
//   * the projection  A^T [Lq | q_mu]  with a fused row-sum-of-squares epilogue (epi = 1)
//
// Design (CDNA4): 256 threads = 4 waves (2x2), workgroup tile BM x BN, wave tile (BM/2) x (BN/2)
// built from v_mfma_f64_16x16x4_f64 (A frag: row = lane&15, k = lane>>4, one f64 per lane; same for
// the B^T frag; D: col = lane&15, row = (lane>>4) + 4*reg).  K is walked in BK=16 slabs, staged
// global -> VGPR -> LDS with register double-buffering and one barrier per slab.  LDS rows are
// padded to 18 doubles (144 B): the 32-lane ds_read_b64 groups then hit 64 distinct banks.
// 2 workgroups/CU (73.7 KB LDS each) keep one wave per SIMD issuing MFMAs while the other waits.
// Block ids are remapped so that (a) each XCD gets a contiguous range of tiles (private L2s) and
// (b) tiles are swept in 8-wide column groups (A/B panel reuse out of the 4 MiB L2).
#include "gpk_internal.h"
#include <mutex>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace {

constexpr int BK = 16;
constexpr int LDSS = BK + 2;
// kernel the launcher picked last (bench profiling facility only; see gpk_profile_gemm_collect_kind):
// 1 gemm_nt_small, 2 + 2 EPI + PAIR gemm_nt_fast<EPI, PAIR>, 6 gemm_nt_kernel
thread_local int g_last_kind = 0;  // (per host thread: the GEMM entry points are reentrant)
constexpr int GROUP_N = 8;

template <int BM, int BN, int WGM, int WGN>
struct TileCfg {
  static constexpr int WM = BM / WGM, WN = BN / WGN;
  static constexpr int TM = WM / 16, TN = WN / 16;
  static constexpr int A_CH = BM * (BK / 2) / 256;
  static constexpr int B_CH = BN * (BK / 2) / 256;
  static constexpr size_t LDS_BYTES = 2 * (size_t)(BM + BN) * LDSS * sizeof(double);
};

// ---- XCD-contiguous + column-grouped tile order -------------------------------------------------
// Tiles are numbered group-of-8-columns major, row-major inside a group; each XCD takes a
// contiguous 1/8 of that sequence.  With c_lower (and square tiles) only the tiles on or below
// the diagonal are numbered, so every XCD gets the same amount of work.
// position nl of the tile sequence -> (tile_m, tile_n)
__device__ __forceinline__ void tile_decode(int nl, int gx, int gy, int compact, int& tile_m, int& tile_n) {
  tile_m = 0; tile_n = 0;
  if (compact) {
    int g = 0;
    for (;; ++g) {
      const int first = g * GROUP_N;
      const int gsz = (gx - first) < GROUP_N ? (gx - first) : GROUP_N;
      const int avail = gy - first;
      const int tr = avail < gsz ? avail : gsz;
      const int cnt = tr * (tr + 1) / 2 + (avail > gsz ? (avail - gsz) * gsz : 0);
      if (nl < cnt) {
        int ro = 0, co = nl;
        const int tri = gsz * (gsz + 1) / 2;
        if (nl < tri) {
          while (co > ro) { co -= ro + 1; ++ro; }
        } else {
          const int w = nl - tri;
          ro = gsz + w / gsz;
          co = w - (w / gsz) * gsz;
        }
        tile_m = first + ro;
        tile_n = first + co;
        break;
      }
      nl -= cnt;
    }
  } else {
    const int gspan = GROUP_N * gy;
    const int group = nl / gspan, within = nl - group * gspan;
    const int first_n = group * GROUP_N;
    const int gsz = (gx - first_n) < GROUP_N ? (gx - first_n) : GROUP_N;
    tile_n = first_n + within % gsz;
    tile_m = within / gsz;
  }
}

__device__ __forceinline__ void tile_order(int lin, int b_tri, int gx, int gy, int total, int compact, int& tile_m,
                                           int& tile_n) {
  const int xcd = lin & 7, local = lin >> 3;
  const int q = total >> 3, r = total & 7;
  int nl = xcd * q + (xcd < r ? xcd : r) + local;
  // triangular-K work (b_tri = 1, many column tiles) is heaviest in the first column groups:
  // keep plain round-robin there so all XCDs walk the groups together, heaviest first
  if (b_tri == 1 && gx > GROUP_N) nl = lin;
  tile_decode(nl, gx, gy, compact, tile_m, tile_n);
}

__device__ __forceinline__ d2 load2(const double* __restrict__ base, long ld, int row, int nrows,
                                    int k, int ke, bool vec_ok) {
  d2 v = {0.0, 0.0};
  if (row < nrows && k < ke) {
    const double* ptr = base + (long)row * ld + k;
    if (vec_ok && k + 1 < ke) {
      v = *reinterpret_cast<const d2*>(ptr);
    } else {
      v.x = ptr[0];
      if (k + 1 < ke) v.y = ptr[1];
    }
  }
  return v;
}

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs p, int gx, int gy, int total, int compact) {
  using Cfg = TileCfg<BM, BN, WGM, WGN>;
  constexpr int WM = Cfg::WM, WN = Cfg::WN, TM = Cfg::TM, TN = Cfg::TN;
  constexpr int A_CH = Cfg::A_CH, B_CH = Cfg::B_CH;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // entry signal of a stream hand-off without queue packets (GemmArgs::sig_ptr, as in gemm_nt_small): "everything queued before
  // this kernel on its stream has completed"
  if (p.sig_ptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)
    __hip_atomic_store(p.sig_ptr, p.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int bz = p.k_off_step ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;   // (see fast_tile)

  int tile_m, tile_n;
  if (BM == 64 && BN == 64 && p.tail_first1 > 0) {
    // the last, partial round of a lower-only launch of the 128 x 128 fast tile (launch_fast, "tail split"): positions
    // tail_first + blockIdx.x / 4 of ITS tile sequence (gx, gy in 128-tiles), each as four 64 x 64 quarters
    int tm, tn;
    tile_decode(p.tail_first1 - 1 + ((int)blockIdx.x >> 2), gx, gy, compact, tm, tn);
    tile_m = 2 * tm + (((int)blockIdx.x >> 1) & 1);
    tile_n = 2 * tn + ((int)blockIdx.x & 1);
  } else if (p.tile_snake) {
    // under-filled triangular-K launches (every workgroup resident at once, column tile 0 the heaviest): workgroups x, x + 256,
    // x + 512, ... land on the same compute unit (round-robin placement), so round 0 takes the heaviest 256 tiles in falling
    // order, round 1 the LIGHTEST 256 in rising order, round 2 the next heaviest, ... -- every CU gets the same K total.
    // (A batch of problems with ONE round each alternates the direction from problem to problem instead.)
    const int r = (int)blockIdx.x >> 8, c = (int)blockIdx.x & 255, R = (int)gridDim.x >> 8;
    const int rr = (r & 1) ? R - 1 - (r >> 1) : (r >> 1);
    if (p.tile_snake == 2) {
      // ... and each XCD (workgroup x runs on XCD x % 8) keeps gy / 8 row tiles to itself: A is read by one L2 only
      const int xcd = c & 7, s = c >> 3, rpx = gy >> 3;
      const int q = rr * 32 + (((r ^ bz) & 1) ? 31 - s : s);
      tile_n = q / rpx;
      tile_m = xcd * rpx + (q - tile_n * rpx);
    } else {
      const int nl = rr * 256 + ((r & 1) ? 255 - c : c);
      if (nl >= total) return;
      tile_n = nl / gy;
      tile_m = nl - tile_n * gy;
    }
  } else {
    tile_order(blockIdx.x, p.b_tri, gx, gy, total, compact, tile_m, tile_n);
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (p.c_lower && n0 > m0 + BM - 1) return;
  if (m0 >= p.m || n0 >= p.n) return;

  const double* __restrict__ A = p.A + (long)bz * p.strideA;
  const double* __restrict__ B = p.B + (long)bz * p.strideB;

  int kb = 0, ke = p.k;
  const int koff = bz * p.k_off_step;   // (K-split of a triangular product: this batch entry holds columns koff .. koff + k of the operands)
  if (p.b_tri && n0 + BN <= p.b_tri_rows) {
    if (p.b_tri == 1) {
      int f = n0 + p.b_tri_off - koff;
      kb = (f > 0 ? f : 0) & ~(BK - 1);
    } else {
      int l = n0 + BN + p.b_tri_off - koff;
      ke = l < p.k ? l : p.k;
    }
  }
  if (p.a_tri == 1) {         // rows m0.. of an upper-triangular A are zero left of column m0
    int f = m0 - koff;
    f = (f > 0 ? f : 0) & ~(BK - 1);
    kb = kb > f ? kb : f;
  } else if (p.a_tri == 2) {  // rows ..m0+BM-1 of a lower-triangular A are zero right of column m0+BM-1
    int l = ((m0 + BM + BK - 1) & ~(BK - 1)) - koff;
    l = l < p.k ? l : p.k;
    ke = ke < l ? ke : l;
  }
  const bool vec_ok = ((p.lda & 1) == 0) && ((p.ldb & 1) == 0) &&
                      ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(B) & 15) == 0);

  constexpr int BUF = (BM + BN) * LDSS;  // doubles per LDS buffer: [A tile | B tile]

  d4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  // per-thread staging slots: chunk c = tid + 256 q  ->  tile row c>>3, k offset (c&7)*2.
  // Rows are clamped (then zero-selected) so the fast path is branch-free: all loads of a slab
  // are issued back to back and drain under the MFMAs of the previous slabs.
  // TWO register sets (round 6, late): the loads of slab s+2 go out at the top of slab s, so a slab's data has two slab times to
  // arrive instead of one.  This kernel's slab is short -- 16 MFMAs per wave for a 64 x 64 tile, 0.43 us -- and its K loop ran at the
  // global-load round trip instead (0.97 us per slab: the 64 x 64 remainders of the capped extra-row updates, K = 512, took 62 / 52 /
  // 35 us per SVGP step; the heaviest tile of a few-row projection walks 128 slabs).  Same arithmetic, same order.
  d2 ra0[A_CH], rb0[B_CH], ra1[A_CH], rb1[B_CH];
  const double* pa[A_CH];
  const double* pb[B_CH];
  bool va[A_CH], vb[B_CH];
#pragma unroll
  for (int q = 0; q < A_CH; ++q) {
    const int c = tid + 256 * q;
    const int row = m0 + (c >> 3);
    va[q] = row < p.m;
    pa[q] = A + (long)(va[q] ? row : p.m - 1) * p.lda + (c & 7) * 2;
  }
#pragma unroll
  for (int q = 0; q < B_CH; ++q) {
    const int c = tid + 256 * q;
    const int row = n0 + (c >> 3);
    vb[q] = row < p.n;
    pb[q] = B + (long)(vb[q] ? row : p.n - 1) * p.ldb + (c & 7) * 2;
  }
  const d2 zero2 = {0.0, 0.0};
  auto gload = [&](int k0, d2* __restrict__ ra, d2* __restrict__ rb) {
    if (vec_ok && k0 + BK <= ke) {  // wave-uniform
#pragma unroll
      for (int q = 0; q < A_CH; ++q) {
        ra[q] = *reinterpret_cast<const d2*>(pa[q] + k0);
      }
#pragma unroll
      for (int q = 0; q < B_CH; ++q) {
        rb[q] = *reinterpret_cast<const d2*>(pb[q] + k0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < A_CH; ++q) {
        const int c = tid + 256 * q;
        ra[q] = load2(A, p.lda, m0 + (c >> 3), p.m, k0 + (c & 7) * 2, ke, false);
      }
#pragma unroll
      for (int q = 0; q < B_CH; ++q) {
        const int c = tid + 256 * q;
        rb[q] = load2(B, p.ldb, n0 + (c >> 3), p.n, k0 + (c & 7) * 2, ke, false);
      }
    }
  };
  auto lstore = [&](int buf, const d2* __restrict__ ra, const d2* __restrict__ rb) {
#pragma unroll
    for (int q = 0; q < A_CH; ++q) {
      const int c = tid + 256 * q;
      *reinterpret_cast<d2*>(&smem[buf * BUF + (c >> 3) * LDSS + (c & 7) * 2]) =
          va[q] ? ra[q] : zero2;
    }
#pragma unroll
    for (int q = 0; q < B_CH; ++q) {
      const int c = tid + 256 * q;
      *reinterpret_cast<d2*>(&smem[buf * BUF + (BM + (c >> 3)) * LDSS + (c & 7) * 2]) =
          vb[q] ? rb[q] : zero2;
    }
  };

  const int nkt = ke > kb ? (ke - kb + BK - 1) / BK : 0;
  if (nkt > 0) {
    gload(kb, ra0, rb0);
    lstore(0, ra0, rb0);
    if (nkt > 1) gload(kb + BK, ra1, rb1);
    __syncthreads();
  }
  const int frag_r = lane & 15, frag_k = lane >> 4;
  // slab j travels in register set j & 1: at slab kt its own set is free again (stored to LDS during slab kt - 1) and takes slab
  // kt + 2; the other set holds slab kt + 1, requested a whole slab ago
  auto step = [&](int kt, d2* __restrict__ fa_, d2* __restrict__ fb_, const d2* __restrict__ na_, const d2* __restrict__ nb_) {
    const int cur = kt & 1;
    if (kt + 2 < nkt) gload(kb + (kt + 2) * BK, fa_, fb_);
    __builtin_amdgcn_sched_barrier(0);   // (the loads go out BEFORE anything waits for the other set: without this the zero-select of lstore is hoisted above them)
    const double* as = smem + cur * BUF + (wm * WM + frag_r) * LDSS + frag_k;
    const double* bs = smem + cur * BUF + (BM + wn * WN + frag_r) * LDSS + frag_k;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[i * 16 * LDSS + kk * 4];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[j * 16 * LDSS + kk * 4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nkt) lstore(cur ^ 1, na_, nb_);
    __syncthreads();
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    step(kt, ra0, rb0, ra1, rb1);
    if (kt + 1 < nkt) step(kt + 1, ra1, rb1, ra0, rb0);
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  const int row_base = m0 + wm * WM + (lane >> 4);
  const int col_base = n0 + wn * WN + (lane & 15);
  if (p.epi == 0) {
    double* __restrict__ C = p.C + (long)bz * p.strideC;
    const double alpha = p.alpha, beta = p.beta;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row_base + i * 16 + 4 * r;
        if (row < p.m) {
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = col_base + j * 16;
            if (col < p.n) {
              double* cp = C + (long)row * p.ldc + col;
              double v = alpha * acc[i][j][r];
              if (beta != 0.0) v += beta * (*cp);
              *cp = v;
            }
          }
        }
      }
  } else {
    double* __restrict__ C2 = p.C2 + (long)bz * p.strideC2;
    double* __restrict__ part = p.part + (long)bz * p.stridePart;
    const double alpha = p.alpha;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row_base + i * 16 + 4 * r;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = col_base + j * 16;
          const double v = alpha * acc[i][j][r];
          if (col < p.sq_cols) {
            s += v * v;
          } else if (row < p.m && col - p.sq_cols < p.c2_cols && col < p.n) {
            C2[(long)row * p.ldc2 + (col - p.sq_cols)] = v;
          }
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        // one partial per 64 columns of the output: tile_n * (BN / 64) + (this wave's 64-column slot within the tile)
        if constexpr (WN >= 64) {
          if ((lane & 15) == 0 && row < p.m) part[(long)(tile_n * (BN / 64) + (wn * WN) / 64) * p.part_ld + row] = s;
        } else {
          // several waves share a 64-column slot: their partial sums meet in LDS (free after the K loop) and are added in wave order
          if ((lane & 15) == 0) smem[wn * BM + (row - m0)] = s;
        }
      }
    if constexpr (WN < 64) {
      constexpr int WPS = 64 / WN;   // waves per slot
      __syncthreads();
      if (wn % WPS == 0 && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row_base + i * 16 + 4 * r;
            double s = smem[wn * BM + (row - m0)];
#pragma unroll
            for (int u = 1; u < WPS; ++u) s += smem[(wn + u) * BM + (row - m0)];
            if (row < p.m) part[(long)(tile_n * (BN / 64) + (wn * WN) / 64) * p.part_ld + row] = s;
          }
      }
    }
  }
}


// =====================================================================================================
// One-round-trip 64 x 64 tile for SHORT-K updates that run beside bulk work (round 6, late): the rest-update of a single-leaf
// panel of the SVGP step is K = 128, beta = 1, lower tiles only -- 10 to 400 tiles of ~1 MFLOP.  On gemm_nt_kernel<64, 64, 4, 1>
// every 16-wide slab is a dependent global-load round trip (8 of them, then the read-modify-write of C: 10 in a row), and while
// the extra-row stream's capped GEMM keeps the memory pipes of 224 compute units full a round trip takes several microseconds:
// 36 tiles took 60 us (profiles/r06_step_timeline.txt), longer than the chain's own 41-us period, and every strip waits for the
// previous rest-update.  Here a thread issues ALL its loads -- eight slabs of A and B (32 x 16 bytes) and its 16 values of C --
// before the first barrier; the slabs then go through the same two 18-KB LDS buffers with the same fragment layout, slab order
// and epilogue arithmetic as the generic kernel (bit-identical results).  36 KB of LDS: fits beside any other workgroup.
// prio: s_setprio of the whole workgroup (its waves share their SIMDs with MFMA-bound waves of the bulk kernel).
__global__ __launch_bounds__(256, 2) void gemm_nt_pre64(GemmArgs p, int gx, int gy, int total, int compact, int prio) {
  constexpr int BM = 64, BN = 64, MAXS = 8;
  constexpr int BUF = (BM + BN) * LDSS;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (p.sig_ptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)
    __hip_atomic_store(p.sig_ptr, p.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (prio == 3) __builtin_amdgcn_s_setprio(3);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // wave w: rows 16 w .. 16 w + 15, all 64 columns
  const int bz = blockIdx.y;
  int tile_m, tile_n;
  tile_order(blockIdx.x, 0, gx, gy, total, compact, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (p.c_lower && n0 > m0 + BM - 1) return;
  if (m0 >= p.m || n0 >= p.n) return;
  const double* __restrict__ A = p.A + (long)bz * p.strideA;
  const double* __restrict__ B = p.B + (long)bz * p.strideB;
  double* __restrict__ C = p.C + (long)bz * p.strideC;
  const int nkt = p.k / BK;   // (the launcher: k a multiple of 16, <= 128)

  // staging slots as in gemm_nt_kernel: chunk c = tid + 256 q -> tile row c >> 3, k offset (c & 7) * 2; rows clamped, then zero-selected
  d2 ra[MAXS][2], rb[MAXS][2];
  const double* pa[2];
  const double* pb[2];
  bool va[2], vb[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = tid + 256 * q;
    const int rowa = m0 + (c >> 3), rowb = n0 + (c >> 3);
    va[q] = rowa < p.m;
    vb[q] = rowb < p.n;
    pa[q] = A + (long)(va[q] ? rowa : p.m - 1) * p.lda + (c & 7) * 2;
    pb[q] = B + (long)(vb[q] ? rowb : p.n - 1) * p.ldb + (c & 7) * 2;
  }
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
    if (s < nkt) {   // (wave-uniform)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        ra[s][q] = *reinterpret_cast<const d2*>(pa[q] + s * BK);
        rb[s][q] = *reinterpret_cast<const d2*>(pb[q] + s * BK);
      }
    }
  // C of this lane's 16 outputs (D layout: col = lane & 15 (+ 16 j), row = (lane >> 4) + 4 r)
  const int row_base = m0 + wave * 16 + (lane >> 4);
  const int col_base = n0 + (lane & 15);
  const double alpha = p.alpha, beta = p.beta;
  double cpre[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = row_base + 4 * r, col = col_base + j * 16;
      cpre[r][j] = (beta != 0.0 && row < p.m && col < p.n) ? C[(long)row * p.ldc + col] : 0.0;
    }

  const d2 zero2 = {0.0, 0.0};
  auto lstore = [&](int buf, const d2 (&xa)[2], const d2 (&xb)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = tid + 256 * q;
      *reinterpret_cast<d2*>(&smem[buf * BUF + (c >> 3) * LDSS + (c & 7) * 2]) = va[q] ? xa[q] : zero2;
      *reinterpret_cast<d2*>(&smem[buf * BUF + (BM + (c >> 3)) * LDSS + (c & 7) * 2]) = vb[q] ? xb[q] : zero2;
    }
  };
  d4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
  if (nkt > 0) {
    lstore(0, ra[0], rb[0]);
    __syncthreads();
  }
  const int frag_r = lane & 15, frag_k = lane >> 4;
#pragma unroll
  for (int kt = 0; kt < MAXS; ++kt)
    if (kt < nkt) {
      const int cur = kt & 1;
      const double* as = smem + cur * BUF + (wave * 16 + frag_r) * LDSS + frag_k;
      const double* bs = smem + cur * BUF + (BM + frag_r) * LDSS + frag_k;
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        const double a = as[kk * 4];
        double b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = bs[j * 16 * LDSS + kk * 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[j], acc[j], 0, 0, 0);
      }
      if (kt + 1 < MAXS && kt + 1 < nkt) lstore(cur ^ 1, ra[kt + 1 < MAXS ? kt + 1 : 0], rb[kt + 1 < MAXS ? kt + 1 : 0]);
      __syncthreads();
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row_base + 4 * r;
    if (row < p.m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = col_base + j * 16;
        if (col < p.n) {
          double v = alpha * acc[j][r];
          if (beta != 0.0) v += beta * cpre[r][j];
          C[(long)row * p.ldc + col] = v;
        }
      }
    }
  }
}

bool pre64_ok(const GemmArgs& a) {
  if (a.epi != 0 || a.k <= 0 || a.k > 128 || (a.k & 15) || a.b_tri || a.a_tri || a.k_off_step || a.tile_snake || a.tail_first1) return false;
  if ((a.lda & 1) || (a.ldb & 1) || (a.strideA & 1) || (a.strideB & 1)) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.B) & 15)) return false;
  return true;
}

int launch_pre64(hipStream_t s, const GemmArgs& a, int prio) {
  constexpr size_t LDS_BYTES = 2 * (size_t)(64 + 64) * LDSS * sizeof(double);
  const int gx = gpk_cdiv(a.n, 64), gy = gpk_cdiv(a.m, 64);
  if (gx <= 0 || gy <= 0) return 0;
  int total = gx * gy, compact = 0;
  if (a.c_lower) {   // (the numbering of launch_cfg: tiles on or below the diagonal, column groups of GROUP_N)
    compact = 1;
    total = 0;
    for (int first = 0; first < gx; first += GROUP_N) {
      const int gsz = (gx - first) < GROUP_N ? (gx - first) : GROUP_N;
      const int avail = gy - first;
      if (avail <= 0) break;
      const int tr = avail < gsz ? avail : gsz;
      total += tr * (tr + 1) / 2 + (avail > gsz ? (avail - gsz) * gsz : 0);
    }
    if (total <= 0) return 0;
  }
  g_last_kind = 6;
  // (A/B, level: few tiles asking for 80 KB of LDS so that they cannot share a compute unit with a capped bulk workgroup and run on the CUs
  //  the cap leaves free -- Cm 1.734 - 1.745 against 1.741 - 1.758 ms, profiles/r06_ab_rest_pre64.log; s_setprio 1 / 3 likewise)
  hipLaunchKernelGGL(gemm_nt_pre64, dim3((unsigned)total, (unsigned)(a.batch > 0 ? a.batch : 1), 1), dim3(256), LDS_BYTES, s, a, gx, gy,
                     total, compact, prio);
  GPK_LAUNCH_CHECK();
  return 0;
}


// =====================================================================================================
// Fast path: 128 x 128 x 16 tiles, every K range a multiple of 16, 16-byte aligned rows.
//
// v_mfma_f64_16x16x4_f64 occupies a SIMD's matrix pipe for 64 cycles (measured: 77.4 TFLOP/s chip-wide
// from ONE wave per SIMD, tools/ubench_f64.hip), so a wave has ~16 issue slots per MFMA for everything
// else and the only way to lose throughput is to let the pipe run dry.  The loop is therefore a
// software pipeline in which no MFMA ever waits for data requested in the same phase:
//   * global -> VGPR loads of slab s+1 are issued at the top of slab s (a full slab = 4096 cycles early),
//   * MFMA fragments are double-buffered in registers: the 8 ds_read_b64 of step kk+1 are issued before
//     the 16 MFMAs of step kk,
//   * the VGPR -> LDS stores of slab s+1 are interleaved one-per-MFMA into step kk=2, the workgroup
//     barrier sits between steps 2 and 3, and step 3 (whose fragments were fetched before the barrier)
//     covers the LDS latency of the first fragments of slab s+1.
// With beta != 0 the accumulators start as (beta/alpha) C, loaded in the prologue next to the first
// slab, so the epilogue is store-only.
// rev != 0: the K slabs are walked from the LAST to the first (same slabs, same per-slab arithmetic; the sum over slabs is
// taken in the opposite order).  Used by the paired triangular-K launches: see gemm_nt_fast.
template <int EPI>
__device__ __forceinline__ void fast_tile(const GemmArgs& p, int tile_m, int tile_n, double* smem, int rev = 0, int bz_queue = -1) {
  constexpr int BM = 128, BN = 128;
  constexpr int BUF = (BM + BN) * LDSS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // (K-split of a triangular product: the LAST chunks hold the most non-empty tiles -- they are dispatched first)
  const int bz = bz_queue >= 0 ? bz_queue : (p.k_off_step ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (p.c_lower && n0 > m0 + BM - 1) return;

  const double* __restrict__ A = p.A + (long)bz * p.strideA;
  const double* __restrict__ B = p.B + (long)bz * p.strideB;
  int kb = 0, ke = p.k;
  const int koff = bz * p.k_off_step;   // (see gemm_nt_kernel)
  if (p.b_tri && n0 + BN <= p.b_tri_rows) {
    if (p.b_tri == 1) {
      const int f = n0 + p.b_tri_off - koff;
      kb = (f > 0 ? f : 0) & ~(BK - 1);
    } else {
      const int l = n0 + BN + p.b_tri_off - koff;
      ke = l < p.k ? l : p.k;
    }
  }
  if (p.a_tri == 1) {         // (see gemm_nt_kernel)
    int f = m0 - koff;
    f = (f > 0 ? f : 0) & ~(BK - 1);
    kb = kb > f ? kb : f;
  } else if (p.a_tri == 2) {
    int l = ((m0 + BM + BK - 1) & ~(BK - 1)) - koff;
    l = l < p.k ? l : p.k;
    ke = ke < l ? ke : l;
  }
  const int nk = ke > kb ? (ke - kb) / BK : 0;

  // ---- staging: thread t moves 16 B of row (t>>3) + 32 q, k offset 2 (t&7), for A and for B -----
  // addresses = wave-uniform 64-bit base (advanced per slab) + per-thread 32-bit byte offset
  const int srow = tid >> 3, scol = (tid & 7) * 2;
  const int mrows = p.m - m0, nrows = p.n - n0;  // rows of this tile that exist (clamp the rest)
  const char* abase = reinterpret_cast<const char*>(A + (long)m0 * p.lda + kb);
  const char* bbase = reinterpret_cast<const char*>(B + (long)n0 * p.ldb + kb);
  unsigned oa[4], ob[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int ra = srow + 32 * q, rb = srow + 32 * q;
    ra = ra < mrows ? ra : mrows - 1;  // clamped rows only feed outputs that are never stored
    rb = rb < nrows ? rb : nrows - 1;
    oa[q] = (unsigned)(((long)ra * p.lda + scol) * 8);
    ob[q] = (unsigned)(((long)rb * p.ldb + scol) * 8);
  }
  const int woff = srow * LDSS + scol;
  d2 st[8];
  auto gload = [&](int s) {
    const long so = rev ? (long)(nk - 1 - s) : (long)s;
    const char* ab = abase + so * (BK * 8);
    const char* bb = bbase + so * (BK * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) st[q] = *reinterpret_cast<const d2*>(ab + oa[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) st[4 + q] = *reinterpret_cast<const d2*>(bb + ob[q]);
  };
  auto lstore1 = [&](int buf, int q) {
    const int row = (q < 4) ? 32 * q : BM + 32 * (q - 4);
    *reinterpret_cast<d2*>(&smem[buf * BUF + row * LDSS + woff]) = st[q];
  };

  d4 acc[4][4];
  // Accumulator layout.  The MFMA's M index (D row = (lane >> 4) + 4 reg) is fed from the B tile and its N index
  // (D column = lane & 15) from the A tile, and the B-tile row that MFMA row x = g + 4 r reads is permuted to
  // 2 g + (r & 1) + 8 (r >> 1).  A lane (c = lane & 15, g = lane >> 4) then owns, of every 16 x 16 block (i, j), row
  // 16 i + c and the COLUMN PAIRS {2 g, 2 g + 1} (registers 0, 1) and {8 + 2 g, 9 + 2 g} (registers 2, 3): the accumulator
  // preload and the store are 16-byte accesses (32 + 32 per thread and tile, the four g lanes of a row covering 64
  // contiguous bytes per instruction) instead of the 64 + 64 8-byte accesses of the plain D layout, which made the
  // prologue / epilogue of the K = 640 trailing updates store-issue-bound (MI355X_MICROARCH.md: 8-byte accesses reach
  // 0.54 - 0.70 of the 16-byte rate).
  const int lane_c = lane & 15, lane_g = lane >> 4;
  const int row_base = m0 + wm * 64 + lane_c;            // + 16 i
  const int col_base = n0 + wn * 64 + 2 * lane_g;        // + 16 j + 8 h (+ 0 / 1)
  // (EPI = 1 with a C operand: the streamed projection's last group squares C + A B^T without storing it)
  const bool load_c = (p.beta != 0.0) && (EPI == 0 || p.C != nullptr);
  if (nk > 0) gload(0);
  // C is addressed as  wave-uniform base + a 32-bit byte offset per accumulator row block (4) + one per column pair (8),
  // both clamped into the matrix
  const int c_r0 = wm * 64 + lane_c, c_c0 = wn * 64 + 2 * lane_g;
  const int c_rmax = p.m - 1 - m0, c_cmax = p.n - 1 - n0;  // last valid row / column of the matrix, relative to the tile
  // (computed where they are used -- prologue and epilogue -- so that no address register lives across the K loop)
  auto c_roff = [&](int i) -> unsigned {
    const int rr = c_r0 + 16 * i;
    return (unsigned)(((long)(rr < c_rmax ? rr : c_rmax) * p.ldc) * 8);
  };
  auto c_coff = [&](int j, int h) -> unsigned {
    const int c = c_c0 + j * 16 + 8 * h;
    return (unsigned)((c < c_cmax ? c : c_cmax) * 8);
  };
  // 16-byte accesses need whole column pairs inside the matrix and 16-byte aligned rows (block-uniform)
  const bool c_vec = (EPI == 0 || p.C != nullptr) && (n0 + BN <= p.n) && !(p.ldc & 1) && !(p.strideC & 1) &&
                     !(reinterpret_cast<uintptr_t>(p.C) & 15);
  if (load_c) {
    const double* __restrict__ C = p.C + (long)bz * p.strideC;
    const double sc = p.beta / p.alpha;
    const char* cb = reinterpret_cast<const char*>(C + (long)m0 * p.ldc + n0);
    if (c_vec) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* rowp = cb + c_roff(i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const d2 lo = *reinterpret_cast<const d2*>(rowp + c_coff(j, 0));
          const d2 hi = *reinterpret_cast<const d2*>(rowp + c_coff(j, 1));
          acc[i][j] = (d4){sc * lo.x, sc * lo.y, sc * hi.x, sc * hi.y};
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* rowp = cb + c_roff(i);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = c_c0 + j * 16 + 8 * (r >> 1) + (r & 1);
            acc[i][j][r] = sc * *reinterpret_cast<const double*>(rowp + (unsigned)((c < c_cmax ? c : c_cmax) * 8));
          }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  }

  // B-tile row read by MFMA row x = lane & 15 (see "Accumulator layout"): a permutation inside each 16-row block, so the
  // 32-lane ds_read_b64 groups still hit 64 distinct banks
  const int bperm = 2 * (lane_c & 3) + ((lane_c >> 2) & 1) + 8 * (lane_c >> 3);
  const double* as = smem + (wm * 64 + lane_c) * LDSS + lane_g;
  const double* bs = smem + (BM + wn * 64 + bperm) * LDSS + lane_g;
  double fa[2][4], fb[2][4];
  auto fload = [&](int buf, int kk, int f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[f][i] = as[buf * BUF + i * 16 * LDSS + kk * 4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[f][j] = bs[buf * BUF + j * 16 * LDSS + kk * 4];
  };
#define GPK_MFMA_ROW(f, i)                                                                         \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] =                                       \
      __builtin_amdgcn_mfma_f64_16x16x4f64(fb[f][j], fa[f][i], acc[i][j], 0, 0, 0)

  if (nk > 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) lstore1(0, q);
    __syncthreads();
    fload(0, 0, 0);
  }
  // one K slab; MORE = another slab follows (its loads / LDS stores / first fragments ride along)
  auto slab = [&](int s, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    const int cur = s & 1;
    // ---- kk = 0 ------------------------------------------------------------------------------
    if constexpr (MORE) gload(s + 1);
    fload(cur, 1, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) { GPK_MFMA_ROW(0, i); }
    if constexpr (MORE) __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);  // 8 global loads
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                      // 8 fragment reads
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                     // 16 MFMA
    __builtin_amdgcn_sched_barrier(0);
    // ---- kk = 1 ------------------------------------------------------------------------------
    fload(cur, 2, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { GPK_MFMA_ROW(1, i); }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- kk = 2: also park slab s+1 in the other LDS buffer, one store per MFMA -------------------
    fload(cur, 3, 1);
    if constexpr (MORE) {
#pragma unroll
      for (int q = 0; q < 8; ++q) lstore1(cur ^ 1, q);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { GPK_MFMA_ROW(0, i); }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    if constexpr (MORE) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 1 DS write
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MORE) __syncthreads();
    // ---- kk = 3 (fragments fetched before the barrier) hides the first reads of slab s+1 -------------
    if constexpr (MORE) fload(cur ^ 1, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { GPK_MFMA_ROW(1, i); }
    if constexpr (MORE) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int s = 0; s + 1 < nk; ++s) slab(s, std::true_type{});
  if (nk > 0) slab(nk - 1, std::false_type{});
#undef GPK_MFMA_ROW

  // ---- epilogue ---------------------------------------------------------------------------------
  if constexpr (EPI == 0) {
    double* __restrict__ C = p.C + (long)bz * p.strideC;
    const double alpha = p.alpha;
    char* cb = reinterpret_cast<char*>(C + (long)m0 * p.ldc + n0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (c_r0 + 16 * i <= c_rmax) {
        char* rowp = cb + c_roff(i);
        if (c_vec) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<d2*>(rowp + c_coff(j, 0)) = (d2){alpha * acc[i][j][0], alpha * acc[i][j][1]};
            *reinterpret_cast<d2*>(rowp + c_coff(j, 1)) = (d2){alpha * acc[i][j][2], alpha * acc[i][j][3]};
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int c = c_c0 + j * 16 + 8 * (r >> 1) + (r & 1);
              if (c <= c_cmax) *reinterpret_cast<double*>(rowp + (unsigned)(c * 8)) = alpha * acc[i][j][r];
            }
        }
      }
    }
  } else {
    double* __restrict__ C2 = p.C2 + (long)bz * p.strideC2;
    double* __restrict__ part = p.part + (long)bz * p.stridePart;
    const double alpha = p.alpha;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row_base + i * 16;
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = col_base + j * 16 + 8 * (r >> 1) + (r & 1);
          const double v = alpha * acc[i][j][r];
          if (col < p.sq_cols) {
            s += v * v;
          } else if (row < p.m && col - p.sq_cols < p.c2_cols && col < p.n) {
            C2[(long)row * p.ldc2 + (col - p.sq_cols)] = v;
          }
        }
      // the four g lanes of a row hold its 64 columns of this wave
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lane_g == 0 && row < p.m) part[(long)(tile_n * 2 + wn) * p.part_ld + row] = s;
    }
  }
}


// pair = 0: one tile per workgroup, XCD-contiguous / column-grouped order.
// pair = 1 (triangular-K operand, b_tri = 1): the K range of column tile j shrinks with j, so a workgroup
// takes column tiles j and gx-1-j back to back -- every workgroup then carries the same number of K slabs
// and the launch finishes together instead of waiting for the full-K tiles.
// QUEUE (round 6, late): the queue form is its own instantiation -- with the queue loops and the static walk in ONE kernel it carried
// three inlined copies of the tile, 25 000 instructions and 241 spilled registers.
template <int EPI, bool PAIR, bool QUEUE = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_fast(GemmArgs p, int gx, int gy, int total, int compact) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (p.sig_ptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)   // entry signal (GemmArgs::sig_ptr)
    __hip_atomic_store(p.sig_ptr, p.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if constexpr (!PAIR) {
    // De-phasing: the two workgroups that share a CU are dispatched together and, with equal tile times, stay in
    // lock-step -- both in their load/store prologue and epilogue at the same moment, when neither feeds the MFMA
    // pipes.  In big launches the second resident set (workgroups 256..511 of the dispatch order) therefore starts
    // half a tile late, once; the offset then persists for the whole kernel.
    if (p.stagger_ticks > 0 && (int)blockIdx.x >= p.stagger_first && (int)blockIdx.x < 2 * p.stagger_first && blockIdx.y == 0) {
      const long long t0 = wall_clock64();
      while (wall_clock64() - t0 < p.stagger_ticks) __builtin_amdgcn_s_sleep(32);
    }
    // (ONE call site of the tile in ONE kernel -- a single loop that either fetches from the queue or walks statically -- was 84 spilled
    //  registers and 2 % SLOWER on every workload: Cm 1.79 - 1.81 against 1.75 - 1.77 ms, GPR C2 30.9 - 31.0 against 30.3 - 30.6 ms, same
    //  box, profiles/r06_ab_single_call_site.log.)
    if constexpr (QUEUE) {
      // Tile QUEUE (round 6): persistent workgroups take (batch entry, tile) pairs from a device counter, last batch entry first.  For
      // launches whose tiles differ widely in K -- the K chunks of a triangular x triangular product: 480 of 1024 pairs non-empty, 8 to
      // 32 slabs each -- a static assignment leaves the launch as long as its most loaded compute unit.  Every pair is computed by exactly
      // one workgroup and written to its own output tile: results do not depend on who took what.
      // (One queue per XCD -- a contiguous eighth of the tile sequence per L2, workgroups helping the other queues once theirs is empty --
      //  was measured and removed: FETCH_SIZE of the first N = 16384 trailing update is 50 % higher with the single queue, but C2 30.85 /
      //  30.99 against 30.72 / 30.73 ms, profiles/r06_ab_gpr_tile_queue.log.)
      volatile int* s_next = reinterpret_cast<volatile int*>(&smem[BK]);   // (the padding of LDS row 0: no tile access touches it)
      const int nbatch = p.batch > 0 ? p.batch : 1;
      const int all = total * nbatch;
      for (;;) {
        if (threadIdx.x == 0) *s_next = (int)((unsigned)atomicAdd(p.queue, 1) - (unsigned)p.queue_base);
        __syncthreads();
        const int t = *s_next;
        __syncthreads();   // (s_next is rewritten, and both LDS buffers refilled, only after everybody has read / finished)
        if (t >= all || t < 0) break;
        const int zq = t / total, tq = t - zq * total;
        int tile_m, tile_n;
        tile_order(tq, p.b_tri, gx, gy, total, compact, tile_m, tile_n);
        fast_tile<EPI>(p, tile_m, tile_n, smem, 0, nbatch - 1 - zq);
      }
      return;
    }
    // gridDim.x < total: persistent workgroups, each walks the tile list with stride gridDim.x
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
      int tile_m, tile_n;
      if (EPI == 1 && p.tile_snake) {
        // a triangular-K projection whose PAIRS would not fill the chip twice (launch_fast): one tile per workgroup, workgroups x and
        // x + total / 2 -- the same compute unit under round-robin placement -- take column tiles j and gx-1-j of one row tile
        const int q = t / gy, half = gx >> 1;
        tile_m = t - q * gy;
        tile_n = q < half ? q : gx - 1 - (q - half);
      } else if (p.k_off_step) {
        // K-split of a triangular product: the chunks of one output tile must not meet on one compute unit (workgroup x of every
        // batch entry lands on about the same CU, and the tiles near the origin are non-empty in EVERY chunk: the launch would last
        // as long as unsplit) -- each chunk walks the tile list from its own offset
        int tt = t + (int)blockIdx.y * (total / (int)gridDim.y);
        if (tt >= total) tt -= total;
        tile_order(tt, p.b_tri, gx, gy, total, compact, tile_m, tile_n);
      } else {
        tile_order(t, p.b_tri, gx, gy, total, compact, tile_m, tile_n);
      }
      fast_tile<EPI>(p, tile_m, tile_n, smem);
      if (t + (int)gridDim.x < total) __syncthreads();  // both LDS buffers are about to be refilled
    }
  } else {
    // The workgroups of one row tile (block ids tile_m + j gy: the same XCD, all resident together) read the same rows of A.
    // Column tile j only needs K >= 128 j, so started at their own first slab they would sit at eight different K offsets
    // and every one of them would pull its rows of A through an L2 that cannot hold them (8 row tiles x 2 MB per XCD):
    // FETCH_SIZE of the projection was 6.8 x its algorithmic bytes.  Time-aligned instead: the first tile of every pair walks
    // K DOWN from the common end (all eight start at the same slab), the second, short one walks UP to it (all eight
    // finish at the same slab), so a slab of A is fetched once per XCD and hit by the other seven.  (b_tri = 1 only.)
    const int align = (EPI == 1 && p.b_tri == 1) ? p.pair_k_align : 0;
    const int tile_m = blockIdx.x % gy, j = blockIdx.x / gy;
    fast_tile<EPI>(p, tile_m, j, smem, align);
    if (gx - 1 - j != j) {
      __syncthreads();  // both LDS buffers are about to be refilled
      fast_tile<EPI>(p, tile_m, gx - 1 - j, smem, 0);
    }
  }
}

// counters of the tile-queue launches: a ring of device words per device, never reset -- a launch of `fetches` fetches (one per tile
// and one failing fetch per workgroup) on a word leaves it at a value the host knows, which is the base of the next launch on that
// word (two launches would have to be 1024 launches apart AND in flight together to meet on a word).  No memset, no packet.
int queue_slot(unsigned fetches, int words, int** out, unsigned* base) {   // words: 1, or 8 (one per XCD, all advancing alike)
  constexpr int kRing = 1024, kMaxDev = 16;
  static std::mutex mu;
  static int* ring[2][kMaxDev] = {};
  static unsigned* value[2][kMaxDev] = {};
  static unsigned next[2][kMaxDev] = {};
  const int w = words == 8 ? 1 : 0;
  int dev = 0;
  GPK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDev) return GPK_E_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(mu);
  if (!ring[w][dev]) {
    GPK_HIP(hipMalloc((void**)&ring[w][dev], sizeof(int) * kRing * words));
    GPK_HIP(hipMemset(ring[w][dev], 0, sizeof(int) * kRing * words));
    value[w][dev] = (unsigned*)calloc(kRing, sizeof(unsigned));
    if (!value[w][dev]) return GPK_E_ARG;
  }
  const unsigned slot = next[w][dev]++ % kRing;
  *out = ring[w][dev] + (size_t)slot * words;
  *base = value[w][dev][slot];
  value[w][dev][slot] += fetches;
  return 0;
}

template <int EPI>
int launch_fast(hipStream_t s, const GemmArgs& a) {
  constexpr size_t LDS_BYTES = 2 * (size_t)256 * LDSS * sizeof(double);
  // (function-local statics: initialised once, thread-safe)
  static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_fast<EPI, false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  // (capped launches ask for more, below)
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_fast<EPI, true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
  GPK_HIP(attr0);
  GPK_HIP(attr1);
  const int gx = gpk_cdiv(a.n, 128), gy = gpk_cdiv(a.m, 128);
  if (gx <= 0 || gy <= 0) return 0;
  int total = gx * gy, compact = 0;
  if (a.c_lower && EPI == 0) {
    compact = 1;
    total = 0;
    for (int first = 0; first < gx; first += GROUP_N) {
      const int gsz = (gx - first) < GROUP_N ? (gx - first) : GROUP_N;
      const int avail = gy - first;
      if (avail <= 0) break;
      const int tr = avail < gsz ? avail : gsz;
      total += tr * (tr + 1) / 2 + (avail > gsz ? (avail - gsz) * gsz : 0);
    }
    if (total <= 0) return 0;
  }
  const unsigned nb = (unsigned)(a.batch > 0 ? a.batch : 1);
  // triangular-K operands (K range shrinking with the column tile for b_tri 1, growing for b_tri 2): paired column
  // tiles.  EPI 0 too (the tri-K GEMMs of the reverse pass, gradients.py: 45 -> 60 TFLOP/s class) unless the launch
  // is lower-only or capped.
  {
    const bool pair_ok = (EPI == 1) ? (a.b_tri == 1)
                                    : ((a.b_tri == 1 || a.b_tri == 2) && !a.c_lower && a.max_wgs == 0 && a.b_tri_off == 0 && a.k_off_step == 0);
    // (round 6) pairs that fill the chip at most once -- C3's projection: 4 x 64 = 256 workgroups, one per CU, whose K loop runs at
    // 79 % alone -- run unpaired instead, heavy and light tile of a pair as TWO workgroups of one CU (88 % together)
    if (EPI == 1 && pair_ok && gx >= 4 && !(gx & 1) && a.b_tri_rows >= a.n && a.max_wgs == 0 &&
        (long)(gx / 2) * gy * nb <= GPK_TUNE(PROJ_UNPAIR_UPTO, 256)) {
      GemmArgs b = a;
      b.tile_snake = 1;
      b.stagger_first = 256;
      b.stagger_ticks = 0;
      g_last_kind = 2 + 2 * EPI;
      hipLaunchKernelGGL((gemm_nt_fast<EPI, false>), dim3((unsigned)total, nb, 1), dim3(256), LDS_BYTES, s, b, gx, gy, total, compact);
      GPK_LAUNCH_CHECK();
      return 0;
    }
    if (pair_ok && gx >= 4 && a.b_tri_rows >= a.n) {
      total = ((gx + 1) / 2) * gy;
      g_last_kind = 2 + 2 * EPI + 1;
      GemmArgs ap = a;
      ap.pair_k_align = GPK_TUNE(PAIR_K_ALIGN, 1);
      hipLaunchKernelGGL((gemm_nt_fast<EPI, true>), dim3((unsigned)total, nb, 1), dim3(256), LDS_BYTES, s, ap, gx, gy,
                         total, compact);
      GPK_LAUNCH_CHECK();
      return 0;
    }
  }
  // Tail split (round 5 EXPERIMENT, off: measured no gain).  The N = 16384 trailing updates take 146 us + 0.414 us per tile
  // (7139 tiles: 3102 us ... 1224 tiles: 653 us), i.e. ~0.7 of a 205-us tile round of ramp and drain per launch, 9 % of the
  // 25 ms those launches sum to.  If that were the partly filled LAST round, running the remainder as 64 x 64 quarters on the
  // generic kernel (four times the workgroups, a quarter of the tile time) would recover most of it; built and measured
  // (profiles/r05_ab_gpr_tail_split.log): 31.4 - 31.6 against 31.25 ms -- the overhead does not depend on the remainder
  // (5459 tiles = 11.006 rounds take 11.45 round times, 4949 = 9.98 rounds 10.3): workgroups drift apart over fifteen rounds
  // and the drain is the same ~0.7 round whatever the tile count.
  int tail_tiles = 0;
  if (EPI == 0 && a.c_lower && a.max_wgs == 0 && nb == 1 && !a.b_tri && !a.a_tri && GPK_TUNE(TAIL_SPLIT, 0)) {
    const int slots = 2 * (a.stagger_first > 0 ? a.stagger_first : 256);
    const int r = total % slots;
    if (total >= 2 * slots && r > 0 && r * 100 <= slots * GPK_TUNE(TAIL_SPLIT_PCT, 50)) tail_tiles = r;
  }
  // The same split DOES pay for the capped launches of the extra-row stream (round 5): 224 persistent workgroups walk 768 / 512 /
  // 256 tiles in 4 / 3 / 2 rounds of ~78 us where 3.43 / 2.29 / 1.14 would do -- a few rounds, no drift, and that stream is the
  // critical path of the SVGP step.  The whole rounds stay on the persistent workgroups; the remainder runs as 64 x 64 quarters
  // on every compute unit, for about a third of a round.
  if (EPI == 0 && !a.c_lower && a.max_wgs > 0 && a.max_wgs < total && nb == 1 && !a.b_tri && !a.a_tri &&
      GPK_TUNE(TAIL_SPLIT_CAPPED, 1)) {
    const int r = total % a.max_wgs;
    if (r > 0 && r * 100 <= a.max_wgs * GPK_TUNE(TAIL_SPLIT_CAPPED_PCT, 60)) tail_tiles = r;
  }
  const int total_all = total;
  total -= tail_tiles;
  unsigned nwg = (unsigned)total;
  if (a.max_wgs > 0 && (unsigned)a.max_wgs < nwg) nwg = (unsigned)a.max_wgs;
  GemmArgs b = a;
  g_last_kind = 2 + 2 * EPI;
  {
    // half a tile in 100 MHz ticks: a 128x128x16 slab costs ~1.7 us per workgroup when two share a CU
    // (A/B, 16384^2 x 512, beta = 1: 60.7 -> 63.0 TFLOP/s; lower-only 55.8 -> 58.8; percent of a half tile, 0 = off)
    const int stagger_on = GPK_TUNE(GEMM_STAGGER, 100);
    if (b.stagger_first <= 0) b.stagger_first = 256;
    b.stagger_ticks = (stagger_on && EPI == 0 && !a.b_tri && total >= 1024 && (nwg == (unsigned)total || nwg >= 2u * (unsigned)b.stagger_first))
                          ? (int)((a.k / 16) * 170 * stagger_on / 200)
                          : 0;
  }
  // A CAPPED launch (persistent workgroups, fewer than compute units x 2) asks for more than half of a CU's LDS, so that no two
  // of its workgroups can share a compute unit.  Without that the dispatcher doubles them up on whatever CUs are free at launch
  // time -- the chain's strip holds 80 - 120 CUs for ~10 us -- and, the tile walk being static, the doubled-up pairs run at half
  // speed for the WHOLE kernel: the first extra-row update of an SVGP step took 318 or 483 us depending on what it was launched
  // beside (profiles/r05_step_timeline_before_extra_row_work.txt, round 5).
  size_t lds_bytes = LDS_BYTES;
  if (EPI == 0 && a.max_wgs > 0 && nwg < (unsigned)total && nb == 1) {
    const int kb = GPK_TUNE(CAP_EXCL_LDS_KB, 84);
    if (kb > 0 && kb <= 160 && (size_t)kb * 1024 > LDS_BYTES) lds_bytes = (size_t)kb * 1024;
  }
  if (EPI == 0 && tail_tiles == 0 && a.max_wgs == 0 &&
      ((a.k_off_step && GPK_TUNE(KSPLIT_QUEUE, 1)) || (a.tile_queue && (long)total * nb > 512))) {
    b.stagger_ticks = 0;
    const long all = (long)total * nb;
    const long qw = a.stagger_first > 0 ? 2L * a.stagger_first : GPK_TUNE(QUEUE_WGS, 512);   // (two per compute unit of the launch stream)
    const unsigned wgs = (unsigned)(all < qw ? all : qw);
    int* q = nullptr;
    unsigned qbase = 0;
    const int rcq = queue_slot((unsigned)all + wgs, 1, &q, &qbase);
    if (rcq) return rcq;
    b.queue = q;
    b.queue_base = (int)qbase;
    if constexpr (EPI == 0) {
      static const hipError_t attrq = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_fast<0, false, true>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
      GPK_HIP(attrq);
      hipLaunchKernelGGL((gemm_nt_fast<0, false, true>), dim3(wgs, 1, 1), dim3(256), LDS_BYTES, s, b, gx, gy, total, compact);
    }
    GPK_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL((gemm_nt_fast<EPI, false>), dim3(nwg, nb, 1), dim3(256), lds_bytes, s, b, gx, gy, total,
                     compact);
  GPK_LAUNCH_CHECK();
  if (tail_tiles > 0) {
    using Cfg = TileCfg<64, 64, 4, 1>;
    static const hipError_t attrt = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<64, 64, 4, 1>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    GPK_HIP(attrt);
    GemmArgs t = a;
    t.tail_first1 = total_all - tail_tiles + 1;
    hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 4, 1>), dim3((unsigned)(4 * tail_tiles), 1, 1), dim3(256), Cfg::LDS_BYTES, s, t, gx,
                       gy, 4 * tail_tiles, compact);
    GPK_LAUNCH_CHECK();
  }
  return 0;
}

// 16-byte aligned rows and K ranges that are multiples of 16 everywhere (per-tile b_tri ranges too)
bool fast_ok(const GemmArgs& a) {
  if (GPK_TUNE(GEMM_NO_FAST, 0)) return false;
  if (a.k <= 0 || (a.k & 15) || (a.b_tri && (a.b_tri_off & 15))) return false;
  if ((a.lda & 1) || (a.ldb & 1) || (a.strideA & 1) || (a.strideB & 1)) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.B) & 15)) return false;
  if (a.epi == 0 && a.beta != 0.0 && a.alpha == 0.0) return false;
  if (a.lda > (1L << 21) || a.ldb > (1L << 21)) return false;  // 32-bit byte offsets inside a tile
  return true;
}


// =====================================================================================================
// Latency path for the short GEMMs on the critical path of the factorisation (panel solve  A21 inv(L11)^T,
// inner updates: K <= 128, a few thousand rows).  There the K loop of the tiled kernels is pure latency
// (every 16-wide slab waits a full global-load round trip), so this kernel stages EVERYTHING at once:
// one workgroup = 16 rows x 128 columns, its A rows and the whole B tile go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: one 1 KiB row per wave instruction, no VGPR staging, padded row stride),
// one barrier, then each of the 8 waves runs its 16x16 output tile over the full K with two
// independent accumulators.  One column tile covers n <= 128, so the in-place solve (C aliases A)
// only overwrites rows the workgroup alone has read.
constexpr int SM_BM = 16, SM_BN = 128, SM_THREADS = 512;

// kparts = 2 (round 6): K is staged in two halves -- (16 + 128) rows of 64 + 2 doubles = 74 KB instead of 146 KB -- so that a
// workgroup of this kernel FITS BESIDE a capped bulk workgroup of the extra-row stream (84 KB, launch_fast) on the same compute
// unit.  With the whole-K image the chain's solve / strip needed compute units free of bulk work: 32 of 256 during the capped
// updates of an SVGP step, i.e. three rounds of ~10 us for 88 workgroups (profiles/r06_step_timeline.txt: 18 - 50 us per
// launch instead of 7.5).  One more staging round trip per launch (~2 us) when the chip is empty, which is why it is a choice of
// the caller (GemmArgs::small_kparts).
__global__ __launch_bounds__(SM_THREADS) void gemm_nt_small(GemmArgs p, int ldk, int kparts) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bz = blockIdx.z;
  const int n0 = blockIdx.x * SM_BN;
  const double* __restrict__ A = p.A + (long)bz * p.strideA;
  const double* __restrict__ B = p.B + (long)bz * p.strideB;
  int kb = 0, ke = p.k;
  if (p.b_tri && n0 + SM_BN <= p.b_tri_rows) {
    if (p.b_tri == 1) {
      const int f = n0 + p.b_tri_off;
      kb = (f > 0 ? f : 0) & ~(BK - 1);
    } else {
      const int l = n0 + SM_BN + p.b_tri_off;
      ke = l < p.k ? l : p.k;
    }
  }
  const int kc = ke > kb ? ke - kb : 0;  // multiple of 16
  double* As = smem;                  // [16][ldk]
  double* Bs = smem + SM_BM * ldk;    // [128][ldk]
  const int r = lane & 15, g = lane >> 4;
  const int col = n0 + wave * 16 + r;
  double* __restrict__ C = p.C + (long)bz * p.strideC;
  // stream hand-offs without queue packets (GemmArgs::sig_ptr / wait_ptr)
  if (p.sig_ptr && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    __hip_atomic_store(p.sig_ptr, p.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (p.wait_ptr) {
    if (tid == 0) {
      const long long t0 = wall_clock64();   // (100 MHz; bounded: a lost hand-off must never hang the device -- 0.5 s, then on)
      bool timed_out = false;
      while ((int)(__hip_atomic_load(p.wait_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - p.wait_val) < 0) {
        if (wall_clock64() - t0 >= 50000000LL) { timed_out = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      // a hand-off that never arrived is an internal error: the factorisation status becomes INT_MAX (gpk.h, "info")
      if (timed_out && p.wait_info) atomicMax(p.wait_info, 0x7fffffff);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the producer's tiles were released by ITS kernel end; drop stale lines
  }
  // gridDim.y < number of 16-row blocks: the workgroup walks the row blocks with stride gridDim.y and keeps its B tile
  // (used when the chain is confined to the reserved compute units: ONE round of workgroups, B staged once per CU)
  const int nmb = (p.m + SM_BM - 1) / SM_BM;
  bool first = true;
  for (int mb = blockIdx.y; mb < nmb; mb += gridDim.y) {
    const int m0 = mb * SM_BM;
    if (p.c_lower && n0 > m0 + SM_BM - 1) continue;  // (workgroup-uniform)
    // ---- this wave's 16x16 output tile: columns n0 + 16 wave .. ------------------------------------------
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    if (p.beta != 0.0) {
      const double sc = p.beta / p.alpha;
      const int cc = col < p.n ? col : p.n - 1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int row = m0 + g + 4 * e;
        row = row < p.m ? row : p.m - 1;
        acc0[e] = sc * C[(long)row * p.ldc + cc];
      }
    }
    const int kch = kc / kparts;   // (kparts == 2: the launcher made sure kc is a multiple of 32)
    for (int part = 0; part < kparts; ++part) {
      const int kbp = kb + part * kch;
      // ---- stage: row q of the 144 (16 A rows; 128 B rows: first pass only unless K is staged in parts), one LDS-DMA
      // instruction each ----------------------------------------------------------------------------------------------
      if (2 * lane < kch) {
        for (int q = wave; q < ((first || kparts > 1) ? SM_BM + SM_BN : SM_BM); q += SM_THREADS / 64) {
          const double* src;
          if (q < SM_BM) {
            int rr = m0 + q;
            rr = rr < p.m ? rr : p.m - 1;
            src = A + (long)rr * p.lda + kbp;
          } else {
            int rr = n0 + q - SM_BM;
            rr = rr < p.n ? rr : p.n - 1;
            src = B + (long)rr * p.ldb + kbp;
          }
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2 * lane),
                                           (__attribute__((address_space(3))) void*)(smem + q * ldk), 16, 0, 0);
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): the LDS-DMA rows of this wave have landed (and the C preload)
      __syncthreads();
      const double* ap = As + r * ldk + g;
      const double* bp = Bs + (wave * 16 + r) * ldk + g;
      const int nkk = kch >> 2;
#pragma unroll 4
      for (int kk = 0; kk < nkk; kk += 2) {
        const double a0 = ap[kk * 4], b0 = bp[kk * 4];
        const double a1 = ap[kk * 4 + 4], b1 = bp[kk * 4 + 4];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
      }
      if (part + 1 < kparts) __syncthreads();  // the image in LDS is about to be replaced by the next part
    }
    first = false;
    if (col < p.n) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = m0 + g + 4 * e;
        if (row < p.m) C[(long)row * p.ldc + col] = p.alpha * (acc0[e] + acc1[e]);
      }
    }
    if (mb + (int)gridDim.y < nmb) __syncthreads();  // the A rows in LDS are about to be replaced
  }
}

// =====================================================================================================
// Fused panel kernel of a single-leaf panel of the latency chain (round 6): the in-place panel solve  S = A21 X^T  AND the strip
//  C[:, next block column] -= S S_top^T  in ONE launch -- the two one-shot launches above, back to back on the panel stream, cost
// 7.3 + 7.7 us, most of it launch ramp, a second staging round trip of the rows a workgroup had just produced, and the drain
// of the first kernel.  A workgroup owns 16 rows as before.  Phase 1 is gemm_nt_small's solve (same staging, same arithmetic:
// two alternating accumulators over K = 128); the solved rows go to global memory AND stay in LDS as the A operand of phase 2.
// Phase 2 needs the solved rows of the NEXT diagonal block (the first `nbw` workgroups' rows) as its B tile: those workgroups
// count themselves in `cnt[0]` once their rows are released; everybody polls it (bounded), then stages the B tile from L2 and
// runs gemm_nt_small's update on it.  The last workgroup through phase 1 (`cnt[1]`) publishes "panel solved" (sig_ptr) -- earlier
// than the strip's entry signal used to.  Workgroups are dispatched in index order, so the producers (indices 0 .. nbw-1)
// are resident before any consumer can occupy a compute unit; the results are bit-identical to the two-launch form.
struct PanelFusedArgs {
  double* P; long lda;            // rows below the leaf of the panel's columns: [m, 128] (in / out)
  const double* X;                // the leaf's block inverse [128, 128], row stride 128
  double* C;                      // the next block column of the same rows: [m, n2]
  int m, n2;                      // rows; columns of the strip (<= 128)
  int* cnt;                       // two zeroed words: producers done, workgroups through phase 1
  int* sig_ptr; int sig_val;      // "panel solved"
  const int* wait_ptr; int wait_val; int* wait_info;   // "previous rest-update done" (before C is touched)
};

__global__ __launch_bounds__(SM_THREADS) void panel_fused_kernel(PanelFusedArgs p) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int K = 128, ldk = K + 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * SM_BM;
  const int nwg = gridDim.x, nbw = (p.n2 + SM_BM - 1) / SM_BM < nwg ? (p.n2 + SM_BM - 1) / SM_BM : nwg;
  double* As = smem;
  double* Bs = smem + SM_BM * ldk;
  // ---- phase 1: S = A21 X^T (X lower triangular: b_tri 2 with one column tile = the whole K) ----------------------------------
  for (int q = wave; q < SM_BM + SM_BN; q += SM_THREADS / 64) {
    const double* src;
    if (q < SM_BM) {
      int rr = m0 + q;
      rr = rr < p.m ? rr : p.m - 1;
      src = p.P + (long)rr * p.lda;
    } else {
      src = p.X + (long)(q - SM_BM) * K;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2 * lane),
                                     (__attribute__((address_space(3))) void*)(smem + q * ldk), 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  const double* ap = As + r * ldk + g;
  const double* bp = Bs + (wave * 16 + r) * ldk + g;
  d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll 4
  for (int kk = 0; kk < K / 4; kk += 2) {
    const double a0 = ap[kk * 4], b0 = bp[kk * 4];
    const double a1 = ap[kk * 4 + 4], b1 = bp[kk * 4 + 4];
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
  }
  // The producers of the B tile (the first nbw workgroups) write their rows THROUGH to memory (agent-scope stores) and count
  // themselves in; everybody else stores normally -- their rows are released by the end of the kernel, and "panel solved" is
  // announced by the entry signal of the next kernel of the panel stream (the next leaf).  (A __threadfence() per workgroup --
  // an L2 write-back each -- made the first version of this kernel 10 us slower than the two launches it replaces.)
  const bool producer = (int)blockIdx.x < nbw;
  double sv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sv[e] = 1.0 * (acc0[e] + acc1[e]);
    const int row = m0 + g + 4 * e;
    if (row < p.m) {
      double* dst = p.P + (long)row * p.lda + wave * 16 + r;
      if (producer) __hip_atomic_store(dst, sv[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *dst = sv[e];
    }
  }
  if (producer) __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): this thread's write-through stores have been acknowledged
  __syncthreads();                       // every wave is done reading As / Bs
#pragma unroll
  for (int e = 0; e < 4; ++e) As[(g + 4 * e) * ldk + wave * 16 + r] = sv[e];   // the solved rows: A operand of phase 2
  __syncthreads();
  if (tid == 0) {
    if (producer) __hip_atomic_fetch_add(p.cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // ---- wait: the B tile's rows are solved; the previous rest-update has left the strip's columns
    const long long t0 = wall_clock64();
    bool timed_out = false;
    while (__hip_atomic_load(p.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nbw) {
      if (wall_clock64() - t0 >= 50000000LL) { timed_out = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (p.wait_ptr) {
      while ((int)(__hip_atomic_load(p.wait_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - p.wait_val) < 0) {
        if (wall_clock64() - t0 >= 50000000LL) { timed_out = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    if (timed_out && p.wait_info) atomicMax(p.wait_info, 0x7fffffff);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  // ---- phase 2: C[rows, 0:n2] -= S S_top^T, lower tiles only (c_lower: a workgroup whose rows lie above the block is skipped) ----
  if (p.n2 <= 0) return;
  for (int q = wave; q < SM_BN; q += SM_THREADS / 64) {
    int rr = q < p.n2 ? q : p.n2 - 1;
    rr = rr < p.m ? rr : p.m - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.P + (long)rr * p.lda + 2 * lane),
                                     (__attribute__((address_space(3))) void*)(Bs + q * ldk), 16, 0, 0);
  }
  const int col = wave * 16 + r;
  d4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = c0;
  {
    const int cc = col < p.n2 ? col : p.n2 - 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int row = m0 + g + 4 * e;
      row = row < p.m ? row : p.m - 1;
      c0[e] = -1.0 * p.C[(long)row * p.lda + cc];     // (beta / alpha) C with alpha = -1, beta = 1, as gemm_nt_small does
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
#pragma unroll 4
  for (int kk = 0; kk < K / 4; kk += 2) {
    const double a0 = ap[kk * 4], b0 = bp[kk * 4];
    const double a1 = ap[kk * 4 + 4], b1 = bp[kk * 4 + 4];
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c1, 0, 0, 0);
  }
  if (col < p.n2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = m0 + g + 4 * e;
      if (row < p.m) p.C[(long)row * p.lda + col] = -1.0 * (c0[e] + c1[e]);
    }
  }
}

}  // namespace

bool gpk_panel_fused_ok(const double* P, long lda, const double* X, int m, int nb, int n2) {
  return nb == 128 && m > 0 && n2 > 0 && n2 <= 128 && !(lda & 1) && !(reinterpret_cast<uintptr_t>(P) & 15) &&
         !(reinterpret_cast<uintptr_t>(X) & 15) && gpk_cdiv(m, SM_BM) <= GPK_TUNE(SMALL_MAX_WGS, 512) && lda <= (1L << 21);
}

int gpk_launch_panel_fused(hipStream_t s, double* P, long lda, const double* X, double* C, int m, int n2, int* cnt, int* sig_ptr,
                           int sig_val, const int* wait_ptr, int wait_val, int* wait_info) {
  constexpr size_t lds = (size_t)(SM_BM + SM_BN) * 130 * sizeof(double);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(panel_fused_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  GPK_HIP(attr);
  PanelFusedArgs a{P, lda, X, C, m, n2, cnt, sig_ptr, sig_val, wait_ptr, wait_val, wait_info};
  hipLaunchKernelGGL(panel_fused_kernel, dim3((unsigned)gpk_cdiv(m, SM_BM)), dim3(SM_THREADS), lds, s, a);
  GPK_LAUNCH_CHECK();
  return 0;
}

namespace {

int launch_small(hipStream_t s, const GemmArgs& a) {
  // K staged in two halves (GemmArgs::small_kparts == 2) when every K range of the launch splits into whole 16-slabs: plain
  // K = 64 / 128 operands, or the single-column-tile triangular solve against a leaf's block inverse (b_tri 2, K range = n <= 128)
  const bool parts2 = a.small_kparts == 2 && !(a.k & 31) && (!a.b_tri || (a.b_tri == 2 && a.n <= SM_BN && a.b_tri_off == 0 && !(a.n & 31)));
  const int kparts = parts2 ? 2 : 1;
  const int ldk = a.k / kparts + 2;
  const size_t lds = (size_t)(SM_BM + SM_BN) * ldk * sizeof(double);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_small),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)((SM_BM + SM_BN) * 130 * sizeof(double)));
  GPK_HIP(attr);
  unsigned gy = (unsigned)gpk_cdiv(a.m, SM_BM);
  const unsigned gxs = (unsigned)gpk_cdiv(a.n, SM_BN);
  if (a.max_wgs > 0 && gy * gxs > (unsigned)a.max_wgs) gy = ((unsigned)a.max_wgs + gxs - 1) / gxs;  // row blocks walked in-kernel
  else if (a.small_loop && a.max_wgs <= 0 && gy * gxs > 512u) gy = (512u + gxs - 1) / gxs;
  dim3 grid(gxs, gy, (unsigned)(a.batch > 0 ? a.batch : 1));
  g_last_kind = 1;
  hipLaunchKernelGGL(gemm_nt_small, grid, dim3(SM_THREADS), lds, s, a, ldk, kparts);
  GPK_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================
// Fused in-group solve of right-hand-side ROWS against a column group of the factor (nb <= 4 leaf blocks):
//     for j = 0 .. nb-1:   S_j = E_j X_j^T                       (X_j = L_jj^-1, the leaf's block inverse)
//                          E_j' -= S_j L_j'j^T   for j' > j       (the rest of the group)
// i.e. exactly the 2 nb - 1 launches of the latency kernel above that the right-looking row solve issues per group
// (4 solves + 3 updates at nb = 4), with the SAME arithmetic per element (two alternating accumulators over K = 128,
// the update accumulated onto -C and negated) -- so the results are bit-identical -- but as ONE launch: a workgroup owns
// 16 rows, keeps their nb x (16 x 128) panel in accumulator registers for the whole group (8 waves x one 16 x 16 tile
// per block), and only the 10 operand tiles X_j / L_j'j stream through LDS.  The extra-row stream of an SVGP step spent
// ~150 us per group in those seven dependent launches (mostly launch ramp and drain on a 256-CU chip); this is one.
struct GroupSolveArgs {
  const double* E; long lde;     // rows to solve, columns of the group start at E (in/out unless Eo differs)
  double* Eo; long ldeo;         // solved rows out (may alias E)
  const double* L; long ldl;     // L[c0, c0]: top-left element of the group's diagonal block
  const double* X;               // block inverses of the group, consecutive [nb][128][128]
  int rows, nb;
  long strideE, strideEo, strideL, strideX;   // batched form (blockIdx.y = problem): element offsets between problems
  int stage_barrier;                          // (A/B build) a workgroup barrier at EVERY pipeline stage of group_solve2_kernel
  int j0, j1;                                 // group_solve2_kernel: leaf blocks [j0, j1) are solved by THIS launch (the blocks before j0 by
                                              // earlier ones); the updated, still unsolved blocks >= j1 go back to E (E == Eo then)
};

__global__ __launch_bounds__(512) void group_solve_kernel(GroupSolveArgs p) {
  constexpr int LDK = 130, NBK = 128;
  {
    const long b = blockIdx.y;
    p.E += b * p.strideE; p.Eo += b * p.strideEo; p.L += b * p.strideL; p.X += b * p.strideX;
  }
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  double* As = smem;               // [16][LDK]
  double* Bs = smem + 16 * LDK;    // [128][LDK]
  // (gridDim.x < number of 16-row slivers: the workgroup walks the slivers with stride gridDim.x -- a cap on the resident
  //  workgroups keeps compute units free for the factorisation's chain, GROUP_SOLVE_MAX_WGS in potrf.hip)
  for (int m0 = blockIdx.x * 16; m0 < p.rows; m0 += gridDim.x * 16) {
  if (m0 != (int)blockIdx.x * 16) __syncthreads();   // the previous sliver's last operand tile is no longer read
  int rowi[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int rr = m0 + g + 4 * e;
    rowi[e] = rr < p.rows ? rr : p.rows - 1;
  }
  const int colw = wave * 16 + r;  // this lane's column inside a 128-block
  d4 c[4];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    if (jb < p.nb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) c[jb][e] = p.E[(long)rowi[e] * p.lde + jb * NBK + colw];
    }
  }
  const double* ap = As + r * LDK + g;
  const double* bp = Bs + (wave * 16 + r) * LDK + g;
  auto stage_b = [&](const double* src, long ld) {  // 128 rows of 128 doubles, one LDS-DMA instruction each
    for (int q = wave; q < NBK; q += 8)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)q * ld + 2 * lane),
                                       (__attribute__((address_space(3))) void*)(Bs + q * LDK), 16, 0, 0);
  };
  auto put_a = [&](const d4& v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) As[(g + 4 * e) * LDK + colw] = v[e];
  };
  auto product = [&](d4& acc0, d4& acc1) {
#pragma unroll 4
    for (int kk = 0; kk < 32; kk += 2) {
      const double a0 = ap[kk * 4], b0 = bp[kk * 4];
      const double a1 = ap[kk * 4 + 4], b1 = bp[kk * 4 + 4];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
    }
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j >= p.nb) break;
    // ---- S_j = E_j X_j^T ------------------------------------------------------------------------------------------
    if (j > 0) __syncthreads();  // previous readers of As / Bs are done
    put_a(c[j]);
    stage_b(p.X + (long)j * NBK * NBK, NBK);
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): this wave's LDS-DMA rows have landed
    __syncthreads();
    d4 s0 = {0.0, 0.0, 0.0, 0.0}, s1 = {0.0, 0.0, 0.0, 0.0};
    product(s0, s1);
    d4 sj;
#pragma unroll
    for (int e = 0; e < 4; ++e) sj[e] = 1.0 * (s0[e] + s1[e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rr = m0 + g + 4 * e;
      if (rr < p.rows) p.Eo[(long)rr * p.ldeo + j * NBK + colw] = sj[e];
    }
    if (j + 1 >= p.nb) break;
    __syncthreads();  // everyone has read E_j / X_j
    put_a(sj);
    // ---- E_j' -= S_j L_j'j^T ----------------------------------------------------------------------------------------
#pragma unroll
    for (int jp = 1; jp < 4; ++jp) {
      if (jp <= j || jp >= p.nb) continue;
      if (jp > j + 1) __syncthreads();  // the previous operand tile is no longer read
      stage_b(p.L + (long)jp * NBK * p.ldl + (long)j * NBK, p.ldl);
      __builtin_amdgcn_s_waitcnt(0x0070);
      __syncthreads();
      d4 u0, u1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int e = 0; e < 4; ++e) u0[e] = -1.0 * c[jp][e];  // (beta / alpha) C with alpha = -1, beta = 1
      product(u0, u1);
#pragma unroll
      for (int e = 0; e < 4; ++e) c[jp][e] = -1.0 * (u0[e] + u1[e]);
    }
  }
  }  // sliver loop
}


// Round 5: the same in-group solve with 32 rows per workgroup and the operand tiles PIPELINED through LDS.
// The kernel above stages each 128 x 128 operand tile whole (133 KB, nothing else fits) and waits for it: 10 exposed L2 round trips
// per 16-row sliver, 512 workgroups at one per compute unit = two rounds, ~111 us per 8192 x 512 group on the extra-row stream --
// which is the critical path of the SVGP step from the fourth panel on (profiles/r05_step_timeline_before_extra_row_work.txt).  Here
//   * a workgroup owns TWO 16-row tiles: every B fragment read from LDS feeds two MFMAs, 256 workgroups = one round at 8192 rows;
//   * the operand tiles of all products of the group form ONE stream of K-quarters (128 rows x 32 K = 32 KB, up to 40 of them)
//     that runs two quarters ahead of the MFMAs through a ring of three LDS buffers, across product boundaries -- their addresses
//     do not depend on any result;
//   * the quarters are unpadded; the 16-byte chunk c of tile row r sits in slot c ^ (r & 15) (the permutation is applied on the
//     GLOBAL address of the LDS-DMA lane), so the 32 lanes of a ds_read_b64 group still hit 64 distinct banks.
// Per element the arithmetic is unchanged (K ascending, two alternating accumulators, the update accumulated onto -C and negated).
constexpr int GS2_LDK = 130;                     // A rows: 128 + 2 doubles
template <int QK, int RING>
constexpr size_t gs2_lds() { return (size_t)(32 * GS2_LDK + RING * 128 * QK) * sizeof(double); }

// QK = K columns per pipeline stage.  32: 40 stages of 16 MFMAs per wave, 132 KB of LDS (a compute unit of its own).
// 16: 80 stages of 8 MFMAs, 80.5 KB -- a workgroup then fits BESIDE one 73.7 KB workgroup of the tiled GEMM, so the in-group
// solve no longer waits for the compute units that the chain's rest-update (launched at the same flag) has just taken.
// s_waitcnt vmcnt(n * DPW) for a wave-uniform n in 0 .. NMAX (the instruction takes an immediate)
template <int DPW, int NMAX>
__device__ __forceinline__ void gs2_wait_vm(int n) {
  if constexpr (NMAX > 0) {
    if (n >= NMAX) {
      constexpr int c = NMAX * DPW;
      static_assert(c < 64, "vmcnt");
      __builtin_amdgcn_s_waitcnt(0x0F70 | (c & 15) | ((c >> 4) << 14));
      return;
    }
    gs2_wait_vm<DPW, NMAX - 1>(n);
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
}

// RING = stage buffers; the operand stream runs RING - 1 stages ahead of the MFMAs.
template <int QK, int RING>
__global__ __launch_bounds__(512) void group_solve2_kernel(GroupSolveArgs p) {
  constexpr int LDK = GS2_LDK, NBK = 128;
  constexpr int AHEAD = RING - 1;
  constexpr int NQ = NBK / QK;            // stages per product
  constexpr int QELEMS = 128 * QK;        // doubles per stage buffer
  constexpr int CH = QK / 2;              // 16-byte chunks per row of a stage
  constexpr int DROWS = 64 / CH;          // rows per LDS-DMA instruction
  constexpr int DPW = (128 / DROWS) / 8;  // LDS-DMA instructions per wave and stage
  static_assert(QK == 32 || QK == 16, "stage width");
  {
    const long b = blockIdx.y;
    p.E += b * p.strideE; p.Eo += b * p.strideEo; p.L += b * p.strideL; p.X += b * p.strideX;
  }
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  double* As = smem;                 // [32][LDK]
  double* Bq = smem + 32 * LDK;      // [RING][128][QK], chunk-swizzled
  const int nb = p.nb, j0 = p.j0, j1 = p.j1;
  int nprod = 0;
  for (int j = j0; j < j1; ++j) nprod += nb - j;
  const int nstages = NQ * nprod;
  // swizzle of a tile row's chunks: QK = 32 -> row & 15 (16 chunks), QK = 16 -> (row >> 1) & 7 (8 chunks, two rows per 64 banks)
  auto swz = [](int row) -> int { return QK == 32 ? (row & 15) : ((row >> 1) & 7); };
  // LDS-DMA of one stage: 64 lanes x 16 bytes = DROWS rows x CH chunks per instruction.  The operand tiles come in issue order
  // (j = j0: X_j0, L_(j0+1)j0, ...; then j0 + 1: ...), tracked by (pj, pjp, pq): block column, block row (pjp == pj: the block
  // inverse X_pj), stage inside the tile.  Per lane only a 32-bit element offset inside the tile, for either row stride.
  const int drow = lane / CH, dslot = lane % CH;
  const int drow0 = wave * DPW * DROWS + drow;   // this lane's row in the wave's first copy; copy i adds i * DROWS
  int issue = 0, pj = j0, pjp = j0, pq = 0;
  auto issue_stage = [&]() {
    if (issue < nstages) {
      const bool isx = pjp == pj;
      const double* src = (isx ? p.X + (long)pj * NBK * NBK : p.L + (long)pjp * NBK * p.ldl + (long)pj * NBK) + pq * QK;
      const unsigned ld = isx ? (unsigned)NBK : (unsigned)p.ldl;   // (rows < 128, ld < 2^21: 32-bit element offsets)
      double* dst = Bq + (issue % RING) * QELEMS;
#pragma unroll
      for (int i = 0; i < DPW; ++i) {
        const int rb = wave * DPW + i;
        const int row = drow0 + i * DROWS;
        const double* gsrc = src + ((unsigned)row * ld + (unsigned)((dslot ^ swz(row)) << 1));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                         (__attribute__((address_space(3))) void*)(dst + rb * 128), 16, 0, 0);
      }
      if (++pq == NQ) {
        pq = 0;
        if (pjp + 1 < nb) ++pjp;
        else { ++pj; pjp = pj; }
      }
    }
    ++issue;
  };
  // fragment addresses: B[row 16 w + r][k = 4 kk + g] of a stage -> chunk 2 kk + (g >> 1), half g & 1
  const int brow = (wave * 16 + r) * QK + (g & 1);
  const int bsw = swz(r), bgh = g >> 1;
  const double* ap = As + r * LDK + g;
  for (int m0 = blockIdx.x * 32; m0 < p.rows; m0 += gridDim.x * 32) {
    if (m0 != (int)blockIdx.x * 32) __syncthreads();   // the previous sliver's buffers are no longer read
    issue = 0; pj = j0; pjp = j0; pq = 0;
    int cs = 0;
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) issue_stage();
    const int colw = wave * 16 + r;
    int rowi[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rr = m0 + 16 * t + g + 4 * e;
        rowi[t][e] = rr < p.rows ? rr : p.rows - 1;
      }
    d4 c[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        if (jb >= j0 && jb < nb) {
#pragma unroll
          for (int e = 0; e < 4; ++e) c[t][jb][e] = p.E[(long)rowi[t][e] * p.lde + jb * NBK + colw];
        }
      }
    auto put_a = [&](const d4& v0, const d4& v1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        As[(g + 4 * e) * LDK + colw] = v0[e];
        As[(16 + g + 4 * e) * LDK + colw] = v1[e];
      }
    };
    // one stage of the current product: acc[t][0] takes the even K groups of four, acc[t][1] the odd ones
    // The B rows a wave reads (tile rows 16 w .. 16 w + 15 = its output columns) are the rows IT copies: the operand stream needs
    // no workgroup barrier at all, only the wave's own vmcnt -- the eight waves drift apart and fill each other's LDS waits.  The
    // A rows are shared: one barrier after each put_a (first stage of a product whose A operand changed).
    auto stage = [&](int q, d4 (&acc)[2][2], bool a_changed) {
      // stage cs has landed when at most the stages behind it are in flight: min(AHEAD - 1, stages left) x DPW of this wave's copies
      {
        const int behind = nstages - 1 - cs < AHEAD - 1 ? nstages - 1 - cs : AHEAD - 1;
        gs2_wait_vm<DPW, AHEAD - 1>(behind);
      }
      asm volatile("" ::: "memory");
      if (a_changed || p.stage_barrier) {
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): this wave's A rows are in LDS
        __builtin_amdgcn_s_barrier();
      }
      issue_stage();   // stage cs + AHEAD replaces stage cs - 1 of this wave's rows, whose fragments it has consumed
      // fragments double-buffered in registers, the three LDS reads of step kk + 1 between the two MFMAs of step kk.  (Measured
      // and not kept: the same pipeline hand-issued three steps deep with counted lgkmcnt waits -- 64.3 against 62.3 us per launch,
      // and rings of 5 / 7 stage buffers -- 72 - 84 us: neither the LDS round trip nor the L2 one is what a sliver waits for; the
      // seven barrier pairs around the changes of the shared A rows and the 20-odd us of launch, load and store are.)
      const double* bq = Bq + (cs % RING) * QELEMS + brow;
      double fb[2], fa0[2], fa1[2];
      auto frag = [&](int kk, int f) {
        fb[f] = bq[((2 * kk + bgh) ^ bsw) << 1];
        fa0[f] = ap[q * QK + kk * 4];
        fa1[f] = ap[16 * LDK + q * QK + kk * 4];
      };
      frag(0, 0);
#pragma unroll
      for (int kk = 0; kk < QK / 4; ++kk) {
        if (kk + 1 < QK / 4) frag(kk + 1, (kk + 1) & 1);
        acc[0][kk & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[kk & 1], fb[kk & 1], acc[0][kk & 1], 0, 0, 0);
        acc[1][kk & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[kk & 1], fb[kk & 1], acc[1][kk & 1], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // 1 MFMA
        if (kk + 1 < QK / 4) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // 3 DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // 1 MFMA
      }
      ++cs;
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < j0) continue;
      if (j >= j1) break;
      // ---- S_j = E_j X_j^T --------------------------------------------------------------------------------------------
      if (j > j0) __syncthreads();   // every wave has finished reading the previous A rows
      put_a(c[0][j], c[1][j]);
      d4 acc[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) { acc[t][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc[t][1] = (d4){0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
      for (int q = 0; q < NQ; ++q) stage(q, acc, q == 0);
      d4 sj[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sj[t][e] = 1.0 * (acc[t][0][e] + acc[t][1][e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rr = m0 + 16 * t + g + 4 * e;
          if (rr < p.rows) p.Eo[(long)rr * p.ldeo + j * NBK + colw] = sj[t][e];
        }
      }
      if (j + 1 >= nb) break;
      __syncthreads();   // everyone has read E_j
      put_a(sj[0], sj[1]);
      // ---- E_j' -= S_j L_j'j^T ----------------------------------------------------------------------------------------
#pragma unroll
      for (int jp = 1; jp < 4; ++jp) {
        if (jp <= j || jp >= nb) continue;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][0][e] = -1.0 * c[t][jp][e];  // (beta / alpha) C with alpha = -1, beta = 1
          acc[t][1] = (d4){0.0, 0.0, 0.0, 0.0};
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) stage(q, acc, q == 0 && jp == j + 1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) c[t][jp][e] = -1.0 * (acc[t][0][e] + acc[t][1][e]);
      }
    }
    // a partial launch hands the updated, unsolved blocks back (in place)
#pragma unroll
    for (int jp = 1; jp < 4; ++jp) {
      if (jp < j1 || jp >= nb) continue;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int rr = m0 + 16 * t + g + 4 * e;
          if (rr < p.rows) p.Eo[(long)rr * p.ldeo + jp * NBK + colw] = c[t][jp][e];
        }
    }
  }  // sliver loop
}

template <int QK, int RING>
int launch_group_solve2(hipStream_t s, const GroupSolveArgs& a, int rows, int batch, int max_wgs) {
  static const hipError_t attr2 = hipFuncSetAttribute(reinterpret_cast<const void*>(group_solve2_kernel<QK, RING>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs2_lds<QK, RING>());
  GPK_HIP(attr2);
  unsigned gx2 = (unsigned)gpk_cdiv(rows, 32);
  if (max_wgs > 0 && gx2 * (unsigned)batch > (unsigned)max_wgs) gx2 = (unsigned)std::max(1, max_wgs / batch);
  constexpr size_t lds = gs2_lds<QK, RING>();
  hipLaunchKernelGGL((group_solve2_kernel<QK, RING>), dim3(gx2, (unsigned)batch), dim3(512), lds, s, a);
  GPK_LAUNCH_CHECK();
  return 0;
}

int launch_group_solve(hipStream_t s, const double* E, long lde, double* Eo, long ldeo, int rows, const double* Lgg, long ldl,
                       const double* X, int nb, int batch, long strideE, long strideEo, long strideL, long strideX, int max_wgs,
                       int j0, int j1) {
  if (rows <= 0) return 0;
  if (j1 < 0) j1 = nb;
  if (j0 < 0 || j0 >= j1 || j1 > nb) return GPK_E_ARG;
  if ((j0 > 0 || j1 < nb) && (E != Eo || lde != ldeo || strideE != strideEo || !GPK_TUNE(GROUP_SOLVE_V2, 1))) return GPK_E_UNSUPPORTED;
  if (batch < 1) batch = 1;
  if (!E || !Eo || !Lgg || !X || nb < 1 || nb > 4) return GPK_E_ARG;
  if ((ldl & 1) || (reinterpret_cast<uintptr_t>(Lgg) & 15) || (reinterpret_cast<uintptr_t>(X) & 15)) return GPK_E_UNSUPPORTED;
  if (batch > 1 && ((strideL & 1) || (strideX & 1))) return GPK_E_UNSUPPORTED;   // (16-byte LDS-DMA of every problem's tiles)
  constexpr size_t LDS = (size_t)(16 + 128) * 130 * sizeof(double);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(group_solve_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
  GPK_HIP(attr);
  GroupSolveArgs a{};
  a.E = E; a.lde = lde; a.Eo = Eo; a.ldeo = ldeo; a.L = Lgg; a.ldl = ldl; a.X = X; a.rows = rows; a.nb = nb;
  a.strideE = strideE; a.strideEo = strideEo; a.strideL = strideL; a.strideX = strideX;
  a.stage_barrier = GPK_TUNE(GS2_STAGE_BARRIER, 0);
  a.j0 = j0; a.j1 = j1;
  // Which kernel: the pipelined one (32 rows per workgroup) runs its 10 block products in ~63 us whatever the row count; the
  // staged one (16 rows) needs ~45 us per ROUND of 256 workgroups (one per CU).  tools/group_solve_probe.py, 512 columns, us:
  //   rows 1024: 41 / 62   2048: 47 / 64   4096: 54 / 66   8192: 101 / 74   (staged / pipelined)
  // so the pipelined kernel takes over where the staged one would need a second round.  (The first version of this switch sent
  // everything to the pipelined kernel: the 1024- / 2048- / 4096-row rank shards of the strong-scaling workload lost 5 / 9 / 5 %.)
  const bool partial = j0 > 0 || j1 < nb;
  const long slivers16 = (long)gpk_cdiv(rows, 16) * batch;
  if (GPK_TUNE(GROUP_SOLVE_V2, 1) && (partial || slivers16 > GPK_TUNE(GROUP_SOLVE_V2_MIN_SLIVERS, 256))) {
    // (stage width 16 -- co-resident with a tiled-GEMM workgroup -- measured 3 % SLOWER on the SVGP step, same box: 2.14 - 2.16 against
    //  2.02 - 2.10 ms, profiles/r05_ab_extra_row_stream.log: twice the barriers, and the wait for compute units was not the larger loss)
    switch (GPK_TUNE(GROUP_SOLVE_QK, 32) * 100 + GPK_TUNE(GROUP_SOLVE_RING, 3)) {
      case 1603: return launch_group_solve2<16, 3>(s, a, rows, batch, max_wgs);
      case 1605: return launch_group_solve2<16, 5>(s, a, rows, batch, max_wgs);
      default: return launch_group_solve2<32, 3>(s, a, rows, batch, max_wgs);
    }
  }
  unsigned gx = (unsigned)gpk_cdiv(rows, 16);
  if (max_wgs > 0 && gx * (unsigned)batch > (unsigned)max_wgs) gx = (unsigned)std::max(1, max_wgs / batch);
  hipLaunchKernelGGL(group_solve_kernel, dim3(gx, (unsigned)batch), dim3(512), LDS, s, a);
  GPK_LAUNCH_CHECK();
  return 0;
}

// small-K latency path: K <= 128 in whole 16-slabs, 16-byte aligned rows, modest row count
bool small_ok(const GemmArgs& a) {
  if (GPK_TUNE(GEMM_NO_SMALL, 0) || a.epi != 0) return false;
  if (a.k <= 0 || a.k > 128 || (a.k & 15) || (a.b_tri && (a.b_tri_off & 15)) || a.k_off_step) return false;
  if ((a.lda & 1) || (a.ldb & 1) || (a.strideA & 1) || (a.strideB & 1)) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.B) & 15)) return false;
  if (a.beta != 0.0 && a.alpha == 0.0) return false;
  const long max_wgs = GPK_TUNE(SMALL_MAX_WGS, 512);
  if (a.small_loop && a.batch <= 1) return true;
  return (long)gpk_cdiv(a.m, SM_BM) * gpk_cdiv(a.n, SM_BN) * (a.batch > 0 ? a.batch : 1) <= max_wgs && a.batch < 65536;
}

template <int BM, int BN, int WGM, int WGN>
int launch_cfg(hipStream_t s, const GemmArgs& a) {
  using Cfg = TileCfg<BM, BN, WGM, WGN>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<BM, BN, WGM, WGN>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
  GPK_HIP(attr);
  const int gx = gpk_cdiv(a.n, BN), gy = gpk_cdiv(a.m, BM);
  if (gx <= 0 || gy <= 0) return 0;
  int total = gx * gy, compact = 0;
  if (a.c_lower && BM == BN && a.epi == 0) {
    compact = 1;
    total = 0;
    for (int first = 0; first < gx; first += GROUP_N) {
      const int gsz = (gx - first) < GROUP_N ? (gx - first) : GROUP_N;
      const int avail = gy - first;
      if (avail <= 0) break;
      const int tr = avail < gsz ? avail : gsz;
      total += tr * (tr + 1) / 2 + (avail > gsz ? (avail - gsz) * gsz : 0);
    }
    if (total <= 0) return 0;
  }
  GemmArgs b = a;
  unsigned gridx = (unsigned)total;
  if (b.tile_snake) {   // (see the kernel: needs whole rounds of 256 workgroups per batch entry, or a single problem)
    const int nbatch = a.batch > 0 ? a.batch : 1;
    if (a.b_tri != 1 || compact || total < 256 || (total == 256 && nbatch < 2)) b.tile_snake = 0;
    else if ((gy & 7) == 0 && (total & 255) == 0) b.tile_snake = 2;
    else if (nbatch == 1) { b.tile_snake = 1; gridx = (unsigned)((total + 255) & ~255); }
    else b.tile_snake = 0;
  }
  dim3 grid(gridx, (unsigned)(a.batch > 0 ? a.batch : 1), 1);
  g_last_kind = 6;
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WGM, WGN>), grid, dim3(256), Cfg::LDS_BYTES, s, b, gx, gy, total,
                     compact);
  GPK_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int gpk_gemm_tiles_n(int n) { return gpk_cdiv(n, 128); }
bool gpk_gemm_takes_latency_kernel(const GemmArgs& a) { return a.m > 0 && a.n > 0 && !a.no_small && small_ok(a); }

int gpk_launch_group_solve(hipStream_t s, const double* E, long lde, double* Eo, long ldeo, int rows, const double* Lgg, long ldl,
                           const double* X, int nb, int batch, long strideE, long strideEo, long strideL, long strideX, int max_wgs,
                           int j0, int j1) {
  return launch_group_solve(s, E, lde, Eo, ldeo, rows, Lgg, ldl, X, nb, batch, strideE, strideEo, strideL, strideX, max_wgs, j0, j1);
}

// ---- optional per-launch timing (bench.py roofline leg): HIP events around every GEMM launch, on the
// stream the kernel is launched on.  Off by default; adds two event records per launch when on. -------
namespace {
struct ProfRec { hipEvent_t e0, e1; double flops; int kind; };
bool g_prof_on = false;
ProfRec* g_prof = nullptr;
int g_prof_n = 0, g_prof_cap = 0;

double algorithmic_flops(const GemmArgs& a) {
  // useful multiply-adds only: lower-trapezoid outputs for c_lower, the non-zero K range for b_tri
  const double m = a.m, n = a.n, k = a.k;
  double outs = m * n;
  if (a.c_lower) outs = (m >= n) ? n * (n + 1) / 2 + (m - n) * n : m * (m + 1) / 2;
  double kk = k;
  if (a.b_tri) {
    const double nn = (a.b_tri_rows < a.n ? a.b_tri_rows : a.n);
    const double tri = (nn <= k) ? nn * (nn + 1) / 2 + nn * (k - nn) : k * (k + 1) / 2;
    return 2.0 * m * (tri + (n - nn) * k) * (a.batch > 0 ? a.batch : 1);
  }
  return 2.0 * outs * kk * (a.batch > 0 ? a.batch : 1);
}
}  // namespace

int gpk_profile_gemm_is_on() { return g_prof_on ? 1 : 0; }

extern "C" void gpk_profile_gemm_enable(int on) {
  g_prof_on = on != 0;
  g_prof_n = 0;
}

// total_ms / launches / flops of the GEMM launches recorded since enable(1) whose ALGORITHMIC flop count is at
// least min_flops (0 = all); synchronises the device.  keep != 0 leaves the records in place for another query.
extern "C" int gpk_profile_gemm_collect_min(double min_flops, int keep, double* total_ms, long* launches,
                                            double* flops) {
  GPK_HIP(hipDeviceSynchronize());
  double ms = 0.0, fl = 0.0;
  long cnt = 0;
  for (int i = 0; i < g_prof_n; ++i) {
    if (g_prof[i].flops < min_flops) continue;
    float t = 0.f;
    GPK_HIP(hipEventElapsedTime(&t, g_prof[i].e0, g_prof[i].e1));
    ms += t;
    fl += g_prof[i].flops;
    ++cnt;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = cnt;
  if (flops) *flops = fl;
  if (!keep) g_prof_n = 0;
  return 0;
}
// The phase spanned by the launches with >= min_flops (first such launch's start .. last such launch's end, by HIP
// events) and the algorithmic flops of EVERY recorded launch issued in between, whatever its stream: the chip-wide
// rate of e.g. the trailing-update phase of a factorisation, where the bulk stream's big GEMMs and the look-ahead
// panel's smaller ones share the machine.  Records are kept.
extern "C" int gpk_profile_gemm_window(double min_flops, double* window_ms, double* flops_all, double* flops_matching,
                                       long* launches_all) {
  GPK_HIP(hipDeviceSynchronize());
  int first = -1, last = -1;
  for (int i = 0; i < g_prof_n; ++i)
    if (g_prof[i].flops >= min_flops) { if (first < 0) first = i; last = i; }
  double fa = 0.0, fm = 0.0;
  float t = 0.f;
  long cnt = 0;
  if (first >= 0) {
    GPK_HIP(hipEventElapsedTime(&t, g_prof[first].e0, g_prof[last].e1));
    for (int i = first; i <= last; ++i) {
      fa += g_prof[i].flops;
      if (g_prof[i].flops >= min_flops) fm += g_prof[i].flops;
      ++cnt;
    }
  }
  if (window_ms) *window_ms = t;
  if (flops_all) *flops_all = fa;
  if (flops_matching) *flops_matching = fm;
  if (launches_all) *launches_all = cnt;
  return 0;
}
// the same, restricted to launches of ONE kernel (kind: 1 gemm_nt_small, 2 gemm_nt_fast<0,false>, 3 <0,true>, 4 <1,false>,
// 5 <1,true>, 6 gemm_nt_kernel) -- directly comparable with rocprofv3's per-kernel average.  Records are kept.
extern "C" int gpk_profile_gemm_collect_kind(int kind, double min_flops, double* total_ms, long* launches, double* flops) {
  GPK_HIP(hipDeviceSynchronize());
  double ms = 0.0, fl = 0.0;
  long cnt = 0;
  for (int i = 0; i < g_prof_n; ++i) {
    if (g_prof[i].kind != kind || g_prof[i].flops < min_flops) continue;
    float t = 0.f;
    GPK_HIP(hipEventElapsedTime(&t, g_prof[i].e0, g_prof[i].e1));
    ms += t;
    fl += g_prof[i].flops;
    ++cnt;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = cnt;
  if (flops) *flops = fl;
  return 0;
}
extern "C" int gpk_profile_gemm_collect(double* total_ms, long* launches, double* flops) {
  return gpk_profile_gemm_collect_min(0.0, 0, total_ms, launches, flops);
}

static int launch_select(hipStream_t s, const GemmArgs& a);

// the same per-launch timing for kernels outside this file (kind 7: the single-launch SVGP step kernel, mega.hip)
// grow the record table of the profiling facility.  The table pointer is published right after realloc (the old block may
// have moved) and the capacity only ever covers records whose two events exist: a failed hipEventCreate leaves a shorter,
// consistent table instead of a dangling pointer (advisor, round 4).
static int prof_grow() {
  const int cap = g_prof_cap ? 2 * g_prof_cap : 1024;
  ProfRec* p = (ProfRec*)realloc(g_prof, sizeof(ProfRec) * cap);
  if (!p) return GPK_E_ARG;
  g_prof = p;
  for (int i = g_prof_cap; i < cap; ++i) {
    if (hipEventCreate(&p[i].e0) != hipSuccess) return i > g_prof_n ? 0 : GPK_E_ARG;
    if (hipEventCreate(&p[i].e1) != hipSuccess) {
      (void)hipEventDestroy(p[i].e0);
      return i > g_prof_n ? 0 : GPK_E_ARG;
    }
    g_prof_cap = i + 1;
  }
  return 0;
}

int gpk_prof_begin(hipStream_t s, double flops, int kind) {
  if (!g_prof_on) return -1;
  if (g_prof_n == g_prof_cap && (prof_grow() != 0 || g_prof_n == g_prof_cap)) return -1;
  ProfRec& r = g_prof[g_prof_n];
  r.flops = flops;
  r.kind = kind;
  if (hipEventRecord(r.e0, s) != hipSuccess) return -1;
  return g_prof_n++;
}
void gpk_prof_end(int idx, hipStream_t s) {
  if (idx >= 0 && idx < g_prof_n) (void)hipEventRecord(g_prof[idx].e1, s);
}

int gpk_launch_gemm(hipStream_t s, const GemmArgs& a) {
  if (a.m <= 0 || a.n <= 0) return 0;
  if (!g_prof_on) return launch_select(s, a);
  if (g_prof_n == g_prof_cap) {
    const int rcg = prof_grow();
    if (rcg) return rcg;
    if (g_prof_n == g_prof_cap) return GPK_E_ARG;
  }
  ProfRec& r = g_prof[g_prof_n++];
  r.flops = algorithmic_flops(a);
  GPK_HIP(hipEventRecord(r.e0, s));
  const int rc = launch_select(s, a);
  r.kind = g_last_kind;
  GPK_HIP(hipEventRecord(r.e1, s));
  return rc;
}

static int launch_select(hipStream_t s, const GemmArgs& a) {
  const long tiles = (long)gpk_cdiv(a.m, 128) * gpk_cdiv(a.n, 128) * (a.batch > 0 ? a.batch : 1);
  if (a.tile64 == 2 && a.epi == 0) return launch_cfg<32, 64, 2, 2>(s, a);
  if (a.tile64 == 3 && a.epi == 0) return launch_cfg<64, 128, 1, 4>(s, a);
  if (a.tile64 && a.epi == 0) {
    if (GPK_TUNE(REST_PRE64, 1) && pre64_ok(a)) return launch_pre64(s, a, GPK_TUNE(REST_PRIO, 0));
    return launch_cfg<64, 64, 4, 1>(s, a);
  }
  if (!a.no_small && small_ok(a)) return launch_small(s, a);  // K <= 128, <= 512 workgroups: the latency path
  if (a.epi == 1 && a.beta != 0.0 && a.C && !fast_ok(a)) return GPK_E_UNSUPPORTED;  // only the fast tile preloads C for epi 1
  if (a.epi == 0 && a.k >= GPK_TUNE(HALF_TILE_KMIN, 1024) && a.m > 64 && a.n > 64 && (a.max_wgs == 0 || a.max_wgs >= tiles)) {
    // under-filled long-K launches (the M^3 triangular products of the reverse pass: 256 tiles of 128 x 128 = ONE
    // workgroup per CU, so the launch lasts as long as its longest tile, 283 us at M = 2048) go to 64 x 128 tiles:
    // twice the workgroups, half the longest tile.  Training step 7.15 -> 6.90 ms (same box, 300; 600: 7.00).
    const long eff = a.c_lower ? tiles / 2 : tiles;
    if (eff < GPK_TUNE(HALF_TILE_BELOW, 300)) return launch_cfg<64, 128, 1, 4>(s, a);
  }
  if (a.epi == 1 && a.b_tri == 1 && !(a.beta != 0.0 && a.C) && a.m > 64) {
    // under-filled projections (a rank's 1024-row shard of a strong-scaled step: 8 row tiles x 8 column pairs = 64
    // workgroups, ONE of them per four CUs, 296 us for 4.3 GFLOP; a CU cannot finish a 128 x 128 x 16 slab in less than
    // 1.7 us however many workgroups it holds): 64 x 64 tiles, unpaired -- sixteen times the workgroups.
    // tools/proj_small_probe.py (profiles/r03_projection_few_rows.txt), paired 128-row tiles / 64 x 128 / 64 x 64:
    // 1024 x 2048: 296 / 194 / 155 us, 300 x 1024 (P = 2): 162 / 98 / 62 us, 2048 x 2048: 306 / 268 / 221 us; from 256 pairs
    // on the paired 128-row tiles win (4096 x 2048: 327 us against 483 us on 64 x 128).
    const long pairs = (long)((gpk_cdiv(a.n, 128) + 1) / 2) * gpk_cdiv(a.m, 128) * (a.batch > 0 ? a.batch : 1);
    if (pairs < GPK_TUNE(PROJ_SMALL_TILE_BELOW, 200)) {
      // (every 64-column partial slot the reduction reads must be written: 64-wide tiles only if they cover the same
      // slots as the 128-wide ones, else 64 x 128 tiles)
      GemmArgs b = a;
      b.tile_snake = GPK_TUNE(PROJ_SNAKE, 1);
      if (gpk_cdiv(a.n, 64) == 2 * gpk_cdiv(a.n, 128)) {
        if (pairs < GPK_TUNE(PROJ_TILE32_BELOW, 100)) return launch_cfg<32, 64, 2, 2>(s, b);
        return launch_cfg<64, 64, 4, 1>(s, b);
      }
      return launch_cfg<64, 128, 2, 2>(s, b);
    }
  }
  if (fast_ok(a) && (a.epi == 1 || (a.n > 64 && (tiles >= 24 || a.m <= 64)))) {
    return a.epi == 1 ? launch_fast<1>(s, a) : launch_fast<0>(s, a);
  }
  if (a.epi == 1) return launch_cfg<128, 128, 2, 2>(s, a);
  if (a.n <= 64) return launch_cfg<128, 64, 2, 2>(s, a);
  // narrow / small problems: 64-row tiles double the number of workgroups (256 CUs to fill)
  const long tiles128 = (long)gpk_cdiv(a.m, 128) * gpk_cdiv(a.n, 128) * (a.batch > 0 ? a.batch : 1);
  if (tiles128 < 192 && a.m > 64) return launch_cfg<64, 128, 1, 4>(s, a);
  return launch_cfg<128, 128, 2, 2>(s, a);
}


extern "C" int gpk_gemm_nt(void* stream, int m, int n, int k, double alpha, const double* A,
                           long lda, const double* B, long ldb, double beta, double* C, long ldc,
                           int b_tri, int c_lower, int batch, long strideA, long strideB,
                           long strideC) {
  if (m < 0 || n < 0 || k < 0 || !A || !B || !C) return GPK_E_ARG;
  GemmArgs g{};
  g.A = A; g.lda = lda; g.strideA = strideA;
  g.B = B; g.ldb = ldb; g.strideB = strideB;
  g.C = C; g.ldc = ldc; g.strideC = strideC;
  g.m = m; g.n = n; g.k = k; g.alpha = alpha; g.beta = beta;
  g.c_lower = c_lower; g.b_tri = b_tri & 3; g.b_tri_off = 0; g.b_tri_rows = n;
  // bit 8: the batch is a K-SPLIT of one triangular product -- entry z holds columns z k .. (z + 1) k of both operands (strided views),
  // and the triangular statements are about the UNSPLIT column index.  The caller sums the `batch` partial products.
  const int ksplit = (b_tri >> 8) & 1;
  g.a_tri = (m <= k * (ksplit ? (batch > 0 ? batch : 1) : 1)) ? ((b_tri >> 4) & 3) : 0;  // (a hint: ignoring it is always correct)
  if ((b_tri & ~0x133) || g.b_tri == 3 || g.a_tri == 3 || (ksplit && (k & 15))) return GPK_E_ARG;
  g.k_off_step = ksplit ? k : 0;
  g.epi = 0; g.batch = batch > 0 ? batch : 1;
  // (A/B: the tile queue for every batched launch -- the split-K products of the reverse pass, 1088 equal tiles -- is level:
  //  training step 5.89 / 6.01 without, 6.07 / 5.95 with it, profiles/r06_ab_train_tri_products.log)
  g.tile_queue = (g.batch > 1 || GPK_TUNE(GEMM_NT_QUEUE_SINGLE, 0)) && GPK_TUNE(GEMM_NT_QUEUE_BATCH, 0);
  if (kGpkExp) g.max_wgs = GPK_TUNE(GEMM_NT_MAX_WGS, 0);   // (A/B build only: tools/capped_gemm_probe.py)
  return gpk_launch_gemm((hipStream_t)stream, g);
}
