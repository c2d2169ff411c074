// Single-launch SVGP step ("step kernel"): the whole dependent part of SVGP.elbo (svgp.py:166-181) -- Cholesky of Kuu
// (conditionals/util.py:67), A^T = Kfu Lm^-T (util.py:125), fmean / sum A^2 (util.py:133,144), the projection onto q_sqrt
// (util.py:151-164) and the Gaussian variational expectations (likelihoods/scalar_continuous.py:139-148) -- as ONE
// persistent kernel with one workgroup per compute unit.
//
// Why.  Rounds 1-3 ran this as ~80 launches on four streams: a latency chain (leaf -> panel solve -> strip, 16 panels)
// beside bulk GEMM launches.  Every variant lost time to the same two things: the chain's kernels need EMPTY
// compute units and wait for the bulk launches to end, and every bulk launch pays ~40 us of ramp and drain.  Here nothing
// is launched while the step runs: a workgroup owns a compute unit for the whole step and interleaves its share of the
// factorisation with its share of the minibatch rows, so the chain never queues behind bulk work and bulk work has no
// launch boundaries.
//
// Work decomposition.  Everything is the right-looking row recurrence of the trapezoidal factorisation
//     for panel q:   X[r, q] = X[r, q] inv(L_qq)^T                      ("FIN":  finish column block q of row block r)
//                    X[r, n] -= X[r, q] L[n, q]^T   for n > q            ("UPD":  update the columns still to come)
// applied to 32-row blocks r of (i) the square part (rows of Kuu below panel q: "chain tasks", which together with the
// 128 x 128 leaf ARE the Cholesky factorisation) and (ii) the minibatch rows of Kfu (each workgroup owns one block for
// the whole step: "bulk"), where it is followed by the projection
//                    C_p[r, n] += X[r, q] Lq_p[q, n]   for n <= q        ("PROJ": right-looking too, so that the work of a
//                                                                         step is the same for every q: M + 128 B-rows)
// One primitive serves all of it: a 32 x 128 A panel resident in LDS, B streamed in slabs of 32 rows (1 KiB rows,
// LDS-DMA, double-buffered), each slab a 32 x 32 x 128 product on four "consumer" waves (one per SIMD: MFMA-bound) while
// the four "producer" waves move the next slab and poll the flags.  Per step and workgroup: 68 slabs.
//
// Scheduling.  Chain tasks of panel p, in dependency order: LEAF(p) | FIN(p+1, 4 quarters) | UPD(p+1, .) | FIN(i>p+1, .)
// | UPD(i>p+1, .), dealt round-robin to the workgroups; a workgroup runs its chain tasks in that global order, polls
// their readiness between (and, through a producer wave, DURING) its bulk streams, and leaves a bulk stream at a slab
// boundary when a task becomes ready.  Every wait is a non-blocking poll of flags in global memory, every task only
// depends on tasks earlier in the global order, so the earliest unfinished task can always run: no deadlock.  All polls
// are bounded by a wall-clock timeout that makes every workgroup leave (info = INT_MAX) instead of hanging the device.
//
// Coherence (8 XCDs, private L2s).  What the minibatch rows read from the factorisation -- the finished rows of L and the
// block inverses -- is WRITE-ONCE data in buffers of its own (Lfin, invd): an address is written (write-through) exactly
// once per step and is never read before that, so after one invalidate at kernel start a plain, cached load can only miss
// and fetch the final value.  No cache is invalidated while the step runs, and the B rows that all 32 workgroups of an
// XCD stream stay in its L2 (first version: two L2 invalidates per workgroup and step, every B row came from the fabric,
// 2.3 - 2.7 us per slab against 0.9 us of MFMA work).  Protocol 0 (reference, slower): producers finish a task with an agent-scope release fence (L2 write-back)
// before raising its flag, consumers run an agent-scope acquire fence (L1 / L2 invalidate) after seeing it -- the
// documented sequences.  Protocol 1: chain data is written with agent-scope write-through stores and read with
// agent-scope loads / LDS-DMA (no L2 flush, no invalidate by chain tasks); bulk workgroups still acquire once per phase.
// Bulk rows (A^T, C) are private to their workgroup: plain accesses.
//
// Numerics: identical recurrence and block inverses as the multi-launch path (gpk_potrf); summation order inside a
// 128-block differs (two alternating accumulators over K = 128), results agree to ~1e-13 relative (tests).
#include "gpk_internal.h"
#include "leaf_device.h"
#include <limits.h>

namespace {

constexpr int NBK = GPK_NB;      // 128: panel width
constexpr int MB = 32;           // rows of a row block (A panel)
constexpr int NS = 32;           // B rows per slab
constexpr int LDP = NBK + 2;     // LDS row stride (doubles): fragment reads of 16 rows x 2 k hit 64 distinct banks
constexpr int PAN = MB * LDP;    // doubles per panel / slab buffer
constexpr int OFF_PA = 0, OFF_PB = PAN, OFF_S0 = 2 * PAN, OFF_S1 = 3 * PAN, OFF_MISC = 4 * PAN;
constexpr int OFF_CS = OFF_MISC + 256;   // three 32 x 32 blocks of old C values (the ring of the flag-synchronised stream)
constexpr int LDS_DOUBLES_MEGA = OFF_CS + 3 * MB * NS;
constexpr int MEGA_THREADS = 512;
constexpr size_t MEGA_LDS = gpk_leaf::LEAF_LDS > (size_t)LDS_DOUBLES_MEGA * 8 ? gpk_leaf::LEAF_LDS : (size_t)LDS_DOUBLES_MEGA * 8;
static_assert(MEGA_LDS + 64 <= 160 * 1024, "LDS budget of one compute unit");

struct MegaArgs {
  double* T; long ld;            // [m + rows, ld]: Kuu (+ jitter) on top, Kfu below (becomes L / A^T in place)
  double* invd;                  // [nb][128][128]
  double* Lfin;                  // [m, ld]: the FINISHED rows of L below the diagonal blocks, each element written exactly once
                                 // (by the FIN task that finishes it) -- what the minibatch rows read as B operand
  const double* LqT; long ldl;   // [P][m][ldl] = tril(q_sqrt_p)^T
  double* Cacc;                  // [P][rows][ld] projection accumulator
  const double* q_mu;            // [m][P]
  const double* Y; long ldy;     // [rows][P]
  double* s0; double* fmean; double* ssq;   // [rows], [rows][P], [P][rows]
  double* partial;               // [nbulk]
  int* flags;
  long long* trace;              // A/B build: [1 + 8 * cap] task trace (count, then records), else NULL
  int trace_wg;                  // A/B build: the workgroup whose bulk quanta are traced too
  long long* stamps;             // [nb][2]: wall clock at the start / publication of every leaf (diagnostics, 16 bytes per panel)
  int* info;
  double* out;
  int m, nb, rows, P, nbulk;
  int rows_pad;                  // rows rounded up to whole row blocks: the caller's T / Cacc have that many minibatch rows
  int one_pool;                  // A/B: deal near and far chain tasks to all workgroups alike
  int dbg;                       // A/B build: what-if bits for the bulk streams (Stream::dbg)
  int role_map;                  // 1: consumer / producer roles from the waves' SIMD ids (default), 0: waves 0..3 / 4..7
  int sync_mode;                 // 0: one workgroup barrier per slab (default), 1: slab hand-over through LDS counters
  double variance, noise, mean_const;
  long long timeout_ticks;
};

// flag words (ints, zeroed before the launch)
__device__ __forceinline__ int f_leaf(int p) { return p; }
__device__ __forceinline__ int f_finc(int nb, int p, int i) { return nb + p * nb + i; }
__device__ __forceinline__ int f_fint(int nb, int p) { return nb + nb * nb + p; }
__device__ __forceinline__ int f_rowd(int nb, int i, int u) { return 2 * nb + nb * nb + 4 * i + u; }
__device__ __forceinline__ int f_done(int nb) { return 6 * nb + nb * nb; }
__device__ __forceinline__ int f_abort(int nb) { return 6 * nb + nb * nb + 1; }

__device__ __forceinline__ int ld_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int add_flag(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef __attribute__((address_space(1))) double gdouble;          // explicitly GLOBAL doubles: inside the non-inlined
typedef __attribute__((address_space(1))) const double cgdouble;   // functions a plain pointer is a flat one, and flat
                                                                    // stores are ordered against every LDS access
// A readiness condition: up to four flag words that must have reached their thresholds.  Always four entries -- unused ones
// point at a valid word with threshold INT_MIN -- so that the test is four unconditional, independent loads and no
// dynamically indexed private array (the first version looped over `n` entries of a private array; the leaf's four-flag
// condition then evaluated true before its flags were set: the leaf ran 10 us ahead of the updates it depends on).
struct Cond { const int* a0; const int* a1; const int* a2; const int* a3; int t0, t1, t2, t3; };
__device__ __forceinline__ void cond_init(Cond& c, const int* valid) {
  c.a0 = c.a1 = c.a2 = c.a3 = valid;
  c.t0 = c.t1 = c.t2 = c.t3 = INT_MIN;
}
__device__ __forceinline__ bool cond_ok(const Cond& c) {
  const int v0 = ld_flag(c.a0), v1 = ld_flag(c.a1), v2 = ld_flag(c.a2), v3 = ld_flag(c.a3);
  return (v0 >= c.t0) & (v1 >= c.t1) & (v2 >= c.t2) & (v3 >= c.t3);
}

template <bool COH>
__device__ __forceinline__ void dma_row(const double* src, double* dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst_wave_uniform, 16, 0, COH ? 16 : 0);
}
// the old C values of a slab: private rows streamed once (non-temporal) unless they are chain data (agent scope)
template <bool COH>
__device__ __forceinline__ void dma_row_nt(const double* src, double* dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst_wave_uniform, 16, 0, COH ? 16 : 2);
}
// C / E traffic of a stream.  COH: agent-scope accesses (chain data shared between XCDs).  Otherwise the rows are private to
// the workgroup and only streamed through once per step: non-temporal, so that 16 KB per slab and workgroup of read-modify-
// write traffic does not push the B rows -- which all 32 workgroups of an XCD read -- out of the 4 MB L2.
template <bool COH>
__device__ __forceinline__ double ld_c(cgdouble* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return __builtin_nontemporal_load(p);
}
template <bool COH>
__device__ __forceinline__ void st_c(gdouble* p, double v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __builtin_nontemporal_store(v, p);
}

// 32 rows x 128 doubles of global memory -> an LDS panel (one LDS-DMA row per wave instruction); rows >= nrows repeat the
// last valid row (their outputs are never stored).  The caller synchronises.
template <bool COH>
__device__ __forceinline__ void load_panel(double* S, int off, const double* src, long ld, int nrows, int wave, int lane) {
  for (int q = wave; q < MB; q += MEGA_THREADS / 64) {
    const int r = q < nrows ? q : nrows - 1;
    dma_row<COH>(src + (long)r * ld + 2 * lane, S + off + q * LDP);
  }
  __builtin_amdgcn_s_waitcnt(0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release fence, i.e. it also
// waits for every outstanding GLOBAL store of the wave -- inside the slab loop that would expose the latency of the C
// stores once per slab (the consumer waves never read those stores back; the end of a task waits for them once).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

enum { MODE_FIN = 0, MODE_SUB = 1, MODE_ADD = 2, MODE_SQ = 3 };
struct Stream {
  const double* B; long ldb;   // B row n at B + n * ldb (128 contiguous doubles = the K segment)
  int n0, n1;                  // B-row range (n0 a multiple of 32)
  double* C; long ldc;         // C[r][n] at C + r * ldc + n, r = 0 .. 31 local rows
  double* C2;                  // MODE_FIN: optional second destination (same ldc): the write-once copy of finished L rows
  int nrows;                   // valid local rows
  int mode;
  int nfresh;                  // MODE_ADD / MODE_SQ: columns n >= nfresh carry no earlier contribution
  int pa_off;                  // LDS offset of the A panel
  long long* tacc;             // A/B build: per-phase cycle counters of the consumer loop (traced workgroup), else NULL
  int dbg;                     // A/B build: what-if bits (results wrong): 1 no C loads, 2 no C stores, 4 no B DMA, 8 no MFMA
  const double* pan_src; long pan_ld;   // if set: the A panel is fetched (32 LDS-DMA rows) in the stream's own prologue,
                                        // together with the first slabs -- one memory round trip instead of two
};

// Runs slabs [slab_begin, nslabs) of a stream; returns the index of the first slab NOT run (== nslabs when complete).  With
// `intr` the producer wave polls that condition while the slabs run and the workgroup leaves at the next slab boundary once
// it holds.  sq: per-lane row sums of squares (MODE_SQ), rows 16 mt + g + 4 r of this wave's column half.
template <bool COH, int MODE>
__device__ __noinline__ int stream_run(const Stream st, int slab_begin, const Cond* intr_in, int* ctl_flat, d4& sq_io, int vw) {
  extern __shared__ __attribute__((aligned(16))) double S[];
  __attribute__((address_space(3))) int* ctl = (__attribute__((address_space(3))) int*)ctl_flat;
  gdouble* Cg = (gdouble*)st.C;
  gdouble* Cg2 = (gdouble*)st.C2;
  // (everything the loops use is copied into registers first: with the descriptors left in memory the compiler orders
  // every LDS-DMA instruction against their reloads and the eight rows of a slab are fetched one after the other)
  const bool has_intr = intr_in != nullptr;
  Cond ic;
  if (has_intr) ic = *intr_in;
  else { ic.a0 = ic.a1 = ic.a2 = ic.a3 = nullptr; ic.t0 = ic.t1 = ic.t2 = ic.t3 = 0; }
  const Cond* intr = has_intr ? &ic : nullptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int pwave = __builtin_amdgcn_readfirstlane(tid >> 6);   // physical wave index: only for splitting the panel rows
  const int wave = __builtin_amdgcn_readfirstlane(vw);          // ROLE index: 0..3 consumers (one per SIMD), 4..7 producers
  const int nslabs = (st.n1 - st.n0 + NS - 1) / NS;
  if (slab_begin >= nslabs) return nslabs;
  const int c = lane & 15, g = lane >> 4;
  const int mt = wave & 1, nt = (wave >> 1) & 1;
  // Slab ring.  Only one of the two panels holds the A operand of a non-FIN stream; the other one serves as a third slab
  // buffer: slabs are then fetched TWO ahead.  Measured with two buffers (one slab ahead): 2.3 us per slab against 0.85 us of MFMA
  // work -- a slab's DMA (L2 miss -> fabric) takes longer than one slab of arithmetic, so the loop ran at memory latency.
  const int nbuf = MODE != MODE_FIN ? 3 : 2;   // (MODE_FIN writes the other panel: OFF_PB is its output)
  const int depth = nbuf - 1;
  const int third = st.pa_off == OFF_PB ? OFF_PA : OFF_PB;
  auto buf_off = [&](int s) -> int { const int b = s % nbuf; return b == 0 ? OFF_S0 : (b == 1 ? OFF_S1 : third); };
  // producer roles: 4, 5, 6 move the slabs (rows lr = role - 4, + 3, + 6, ...: 11 / 11 / 10 rows), 7 only polls
  const int nd = wave == 6 ? 10 : 11;   // LDS-DMA instructions of this wave per slab
  auto dma_slab = [&](int s) {
    const int off = buf_off(s);
    for (int lr = wave - 4; lr < NS; lr += 3) {
      int n = st.n0 + s * NS + lr;
      n = n < st.n1 ? n : st.n1 - 1;
      dma_row<COH>(st.B + (long)n * st.ldb + 2 * lane, S + off + lr * LDP);
    }
  };
  if (tid == 0) { ctl[1] = 0; ctl[2] = 0; }
  if (st.pan_src) {   // the A panel: 4 rows per wave
    for (int q = pwave; q < MB; q += MEGA_THREADS / 64) {
      const int r = q < st.nrows ? q : st.nrows - 1;
      dma_row<COH>(st.pan_src + (long)r * st.pan_ld + 2 * lane, S + st.pa_off + q * LDP);
    }
  }
  if (wave >= 4 && wave <= 6) {
    for (int d = 0; d < depth; ++d)
      if (slab_begin + d < nslabs) dma_slab(slab_begin + d);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  // Three loops with the same trip count and one barrier per slab: the poller's, the movers' and the consumers'.
  if (wave == 7) {
    // The poller.  A flag read is a round trip to memory (~2 us): waiting for one per slab made EVERY slab 2 us long (all
    // eight waves meet at the slab barrier).  So: one set of four reads every third slab, tested two slabs later -- this
    // wave has nothing else in flight, the wait the compiler puts in front of the test only covers those reads.
    int v0 = INT_MIN, v1 = INT_MIN, v2 = INT_MIN, v3 = INT_MIN;
    for (int s = slab_begin; s < nslabs; ++s) {
      const int ph3 = (s - slab_begin) % 3;
      int ready = 0;
      if (intr && lane == 0) {
        if (ph3 == 0) { v0 = ld_flag(intr->a0); v1 = ld_flag(intr->a1); v2 = ld_flag(intr->a2); v3 = ld_flag(intr->a3); }
        if (ph3 == 2) ready = ((v0 >= intr->t0) & (v1 >= intr->t1) & (v2 >= intr->t2) & (v3 >= intr->t3)) ? 1 : 0;
        ctl[1 + (s & 1)] = ready;
      }
      lds_barrier();
      if (intr && s + 1 < nslabs && ctl[1 + (s & 1)]) return s + 1;
    }
    if constexpr (MODE == MODE_FIN) lds_barrier();
    return nslabs;
  }
  if (wave >= 4) {
    for (int s = slab_begin; s < nslabs; ++s) {
      const bool more = s + depth < nslabs;
      if (more && !(st.dbg & 4)) dma_slab(s + depth);
      // slab s + 1 must have landed before the barrier; the slab just requested (nd instructions of this wave) may stay in
      // flight when the ring is three deep
      if (more && depth == 2 && !(st.dbg & 4)) {
        if (nd == 11) __builtin_amdgcn_s_waitcnt(0x0F7B);   // vmcnt(11)
        else __builtin_amdgcn_s_waitcnt(0x0F7A);            // vmcnt(10)
      } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
      }
      lds_barrier();
      if (intr && s + 1 < nslabs && ctl[1 + (s & 1)]) {
        __builtin_amdgcn_s_waitcnt(0);   // (nothing of this stream may still be landing in LDS when the caller reuses it)
        return s + 1;
      }
    }
    if constexpr (MODE == MODE_FIN) lds_barrier();
    return nslabs;
  }
  double fa[NBK / 4];   // this lane's A fragments (row 16 mt + c, k = 4 kk + g), fixed for the whole stream
  {
    const double* pa = S + st.pa_off + (16 * mt + c) * LDP + g;
#pragma unroll
    for (int kk = 0; kk < NBK / 4; ++kk) fa[kk] = pa[4 * kk];
  }
  d4 sq = sq_io;
  long long t_top = 0, acc_pre = 0, acc_mma = 0, acc_post = 0, acc_bar = 0;   // A/B build: where a slab's time goes (role 0)
  // Everything per-lane that does not change from slab to slab is computed once: the four row pointers of this lane's C
  // elements (rows 16 mt + g + 4 r, column 16 nt + c of the slab), advanced by NS columns per slab, and the rows' validity.
  // (First version: 64-bit multiply-adds and four-way mode branches per element and slab -- 800 + 1100 cycles of VALU work
  // around 2300 cycles of MFMAs, profiles/r04_mega_consumer_cycles.txt.)  The stream's B-row range is whole slabs.
  gdouble* cp[4];
  gdouble* cp2[4];
  bool rv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = 16 * mt + g + 4 * r;
    rv[r] = rr < st.nrows;
    const long off = (long)(rv[r] ? rr : 0) * st.ldc + st.n0 + (long)slab_begin * NS + 16 * nt + c;
    cp[r] = Cg + off;
    cp2[r] = Cg2 ? Cg2 + off : nullptr;
  }
  const int s_fresh = (MODE == MODE_SUB) ? nslabs : (st.nfresh - st.n0) / NS;   // slabs >= s_fresh carry no earlier contribution
  // C values are read CD slabs ahead of their use (an HBM round trip under a chip-wide read-modify-write stream is longer
  // than one slab of MFMAs)
  constexpr int CD = 3;
  double cq[CD][4];
  auto load_cold = [&](int s, int ahead, double* cold) {
    const bool has_old = (MODE != MODE_FIN) && s < s_fresh && s < nslabs && !(st.dbg & 1);   // (wave-uniform)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cold[r] = 0.0;
      if (has_old && rv[r]) cold[r] = ld_c<COH>(cp[r] + ahead * NS);
    }
  };
#pragma unroll
  for (int d = 0; d < CD; ++d) load_cold(slab_begin + d, d, cq[d]);
  const int pb_lane = (16 * nt + c) * LDP + g;
  // The epilogue of slab s - 1 (adds, stores, pointer bumps) is issued AFTER the first fragment reads of slab s: it then runs
  // in the shadow of that LDS round trip and of the first MFMAs instead of between two slabs.
  double pend[4] = {0.0, 0.0, 0.0, 0.0};   // results of the previous slab, not stored yet
  bool have_pend = false;
  int pend_s = 0;
  auto epilogue = [&]() {
    if constexpr (MODE == MODE_SQ) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sq[r] += pend[r] * pend[r];
    } else {
      if constexpr (MODE == MODE_FIN) {
        const int nl = pend_s * NS + 16 * nt + c;   // column inside the 128-wide output panel
#pragma unroll
        for (int r = 0; r < 4; ++r) S[OFF_PB + (16 * mt + g + 4 * r) * LDP + nl] = pend[r];
      }
      if (!(st.dbg & 2)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (rv[r]) st_c<COH>(cp[r], pend[r]);
        if constexpr (MODE == MODE_FIN) {
          if (Cg2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (rv[r]) st_c<COH>(cp2[r], pend[r]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { cp[r] += NS; if constexpr (MODE == MODE_FIN) cp2[r] += NS; }
  };
  for (int s = slab_begin; s < nslabs; ++s) {
    if (kGpkExp && st.tacc) t_top = __builtin_readcyclecounter();
    // B fragments of the slab in four groups of eight k-steps, group G + 1 in flight under the MFMAs of group G (LDS
    // returns in order, so the wait before a group only covers that group); the A fragments live in registers
    const double* pb = S + buf_off(s) + pb_lane;
    double fb[2][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) fb[0][k] = pb[4 * k];
    __builtin_amdgcn_sched_barrier(0);
    if (have_pend) epilogue();          // (cp points at slab s - 1 until here)
    double cold[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cold[r] = cq[0][r];
#pragma unroll
    for (int d = 0; d + 1 < CD; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) cq[d][r] = cq[d + 1][r];
    load_cold(s + CD, CD, cq[CD - 1]);   // (cp now points at slab s)
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    long long t_a = 0, t_b = 0, t_c = 0;
    if (kGpkExp && st.tacc) { __builtin_amdgcn_sched_barrier(0); t_a = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
      if (grp + 1 < 4) {
#pragma unroll
        for (int k = 0; k < 8; ++k) fb[(grp + 1) & 1][k] = pb[4 * (8 * (grp + 1) + k)];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(st.dbg & 8)) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[8 * grp + k], fb[grp & 1][k], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[8 * grp + k + 1], fb[grp & 1][k + 1], acc1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kGpkExp && st.tacc) { __builtin_amdgcn_sched_barrier(0); t_b = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double o = acc0[r] + acc1[r];
      pend[r] = MODE == MODE_FIN ? o : (MODE == MODE_SUB ? cold[r] - o : cold[r] + o);
    }
    have_pend = true;
    pend_s = s;
    if (kGpkExp && st.tacc) { __builtin_amdgcn_sched_barrier(0); t_c = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
    lds_barrier();
    if (kGpkExp && st.tacc) {
      const long long t_d = __builtin_readcyclecounter();
      acc_pre += t_a - t_top; acc_mma += t_b - t_a; acc_post += t_c - t_b; acc_bar += t_d - t_c;
      if (tid == 0) { st.tacc[0] += acc_pre; st.tacc[1] += acc_mma; st.tacc[2] += acc_post; st.tacc[3] += acc_bar; st.tacc[4] += 1; }
      acc_pre = acc_mma = acc_post = acc_bar = 0;
    }
    if (intr && s + 1 < nslabs && ctl[1 + (s & 1)]) {
      epilogue();
      sq_io = sq;
      return s + 1;
    }
  }
  if (have_pend) epilogue();
  if constexpr (MODE == MODE_FIN) lds_barrier();   // the last slab's part of the output panel (OFF_PB) is in LDS for everybody
  sq_io = sq;
  return nslabs;
}

// ---- the same stream without a workgroup barrier per slab (A/B: GPK_MEGA_SYNC=1; correct, and SLOWER: 3.28 against 3.10 ms) ----
// The barrier version above costs ~3400 consumer cycles per slab of which 2300 are MFMAs (profiles/r04_mega_task_trace_and_
// consumer_cycles.txt): after the slab barrier every consumer first waits for its fragment reads (676 cycles), and the barrier
// itself collects the skew of eight waves (322 cycles).  Here the slab hand-over goes through counters in LDS instead:
//   c_land[j] slabs of which mover wave j's rows are in LDS                     (slab r landed:  every c_land[j] >= r + 1)
//   c_read[i] slabs of which consumer wave i has ISSUED all fragment reads (LDS executes a wave's operations in order, so the
//             store is performed after the reads)                              (slab r free:    every c_read[i] >= r + 1)
//             -- one word per wave: a sum could reach the threshold while one wave is still behind
//   c_issue   number of slabs the LEAD mover has decided to fetch; the other movers follow it
//   c_stop_at first slab that is NOT run: written once by the lead mover when the polling wave has raised c_stop_req -- the
//             lead mover is the only wave that decides where a stream ends, so the four consumers agree on it by construction
// A consumer only blocks when the next slab has not landed; in the steady state it would find it landed while its last MFMA
// group of the current slab is still to be issued and load the first fragments of the next slab under those MFMAs.  Every spin
// loop is bounded (c_abort): the stream then returns -1 and the step fails with info = INT_MAX instead of hanging the device.
//
// What the measurement says (profiles/r04_mega_flag_sync_whatif.txt): the instruction stream is what it should be -- no wait on an
// MFMA result, no global load in the consumers, every LDS wait three MFMA pairs behind its request -- and a slab still takes 4080
// cycles, 1216 of them in the blocking wait: 1002 of 1088 slabs are NOT in LDS when they are needed.  With the B-slab DMA
// switched off 3088 cycles, with the old-C DMA off 3397, with both off 2868: the stream runs at the latency of its LDS-DMA
// (the first workgroup of an XCD to touch a B row fetches it from memory, ~2 - 3 us; the ring holds two slabs ahead = ~2 us of
// cover) and a deeper ring does not fit: four 33-KB slab buffers + four 8-KB old-C blocks are 166 KB.  The barrier version hides
// part of that latency behind its own inefficiency.  Kept as the A/B variant it is; not the way to 1.8 ms.
typedef __attribute__((address_space(3))) int lint;
__device__ __forceinline__ int lds_ld(lint* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(lint* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_add(lint* p, int v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// A store the compiler does not see as an LDS access: behind an LDS-DMA instruction it orders every DS instruction after ALL
// outstanding DMA (vmcnt(0)) -- the movers publish "slab t - 1 has landed" while slab t is still in flight.
__device__ __forceinline__ void lds_st_raw(lint* p, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void cfence() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }   // compiler-only ordering
enum { C_STOPREQ = 3, C_STOPAT = 4, C_ISSUE = 5, C_FIN = 6, C_ABORT = 7, C_READ0 = 8, C_LAND0 = 12 };   // (ctl has 16 words)
constexpr int NMOV = 3;
constexpr long long SPIN_TICKS = 50000000LL;   // 0.5 s of the 100 MHz clock

#define SB() __builtin_amdgcn_sched_barrier(0)
struct TagTrue { static constexpr bool value = true; };
struct TagFalse { static constexpr bool value = false; };
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// All row blocks are FULL here (the fused driver pads the minibatch rows of its workspace to a multiple of 32; the padding rows
// carry garbage through the arithmetic and are left out of the final sum), so the loop has no per-row predicates and everything
// that controls it is wave-uniform (scalar branches).
//
// The OLD C values of a read-modify-write stream travel with the B slab: the movers fetch the slab's 32 x 32 C block by LDS-DMA
// into a ring of its own (OFF_CS) and the consumers read it from LDS.  A consumer wave therefore issues no global LOADS at all,
// only stores -- which matters because loads and stores share one counter (vmcnt) and complete out of order with respect to
// each other: with both in flight the compiler has to wait for ALL of them before it may use a loaded value, i.e. every slab
// waited for the stores of the previous one (part of the 676 cycles "before the MFMAs" of the barrier version).
template <bool COH, int MODE>
__device__ __noinline__ int stream_run_flags(const Stream st, int slab_begin_in, const Cond* intr_in, int* ctl_flat, d4& sq_io, int vw) {
  extern __shared__ __attribute__((aligned(16))) double S[];
  lint* ctl = (lint*)ctl_flat;
  gdouble* Cg = (gdouble*)st.C;
  gdouble* Cg2 = (gdouble*)st.C2;
  const bool has_intr = intr_in != nullptr;
  Cond ic;
  if (has_intr) ic = *intr_in;
  else { ic.a0 = ic.a1 = ic.a2 = ic.a3 = nullptr; ic.t0 = ic.t1 = ic.t2 = ic.t3 = 0; }
  const int tid = threadIdx.x, lane = tid & 63;
  const int pwave = rfl(tid >> 6);
  const int wave = rfl(vw);          // ROLE: 0..3 consumers (one per SIMD), 4..6 movers (4 leads), 7 poller
  const int slab_begin = rfl(slab_begin_in);
  const int n0 = rfl(st.n0), n1 = rfl(st.n1), pa_off = rfl(st.pa_off);
  const int nslabs = (n1 - n0 + NS - 1) / NS;
  if (slab_begin >= nslabs) return nslabs;
  const int c = lane & 15, g = lane >> 4;
  const int mt = wave & 1, nt = (wave >> 1) & 1;
  const int nbuf = MODE != MODE_FIN ? 3 : 2;
  const int depth = nbuf - 1;
  const int third = pa_off == OFF_PB ? OFF_PA : OFF_PB;
  auto buf_off = [&](int s) -> int { const int b = s % nbuf; return b == 0 ? OFF_S0 : (b == 1 ? OFF_S1 : third); };
  auto cbuf_off = [&](int s) -> int { return OFF_CS + (s % 3) * (MB * NS); };
  const int s_fresh = (MODE == MODE_FIN) ? 0 : ((MODE == MODE_SUB) ? nslabs : (rfl(st.nfresh) - n0) / NS);   // slabs >= s_fresh: no old C
  // mover wave j = wave - 4 moves B rows j, j + 3, ... (11 / 11 / 10 LDS-DMA instructions) and C row quads j, j + 3, ... of 8 (3 / 3 / 2)
  const int dbg = kGpkExp ? rfl(st.dbg) : 0;   // A/B build what-if bits (results wrong): 1 no old-C DMA, 2 no C stores, 4 no B DMA
  auto dma_slab = [&](int s) {
    if (dbg & 4) return;
    const int off = buf_off(s);
    const double* src = st.B + (long)(n0 + s * NS + (wave - 4)) * st.ldb + 2 * lane;
    for (int lr = wave - 4; lr < NS; lr += 3) {
      dma_row<COH>(src, S + off + lr * LDP);
      src += 3 * st.ldb;
    }
  };
  auto dma_cslab = [&](int s) {   // lane l of quad i: row 4 i + (l >> 4), columns 2 (l & 15) .. + 1 of the slab
    if (dbg & 1) return;
    const int off = cbuf_off(s);
    const double* src = st.C + (long)(4 * (wave - 4) + (lane >> 4)) * st.ldc + n0 + s * NS + 2 * (lane & 15);
    for (int i = wave - 4; i < MB / 4; i += 3) {
      dma_row_nt<COH>(src, S + off + i * (4 * NS));
      src += 12 * st.ldc;
    }
  };
  const int npre = nslabs - slab_begin < depth ? nslabs - slab_begin : depth;   // slabs fetched in the prologue
  if (tid == 0) {
    for (int j = 0; j < NMOV; ++j) lds_st(ctl + C_LAND0 + j, npre);
    for (int i = 0; i < 4; ++i) lds_st(ctl + C_READ0 + i, 0);
    lds_st(ctl + C_STOPREQ, 0); lds_st(ctl + C_STOPAT, INT_MAX);
    lds_st(ctl + C_ISSUE, slab_begin + npre); lds_st(ctl + C_FIN, 0); lds_st(ctl + C_ABORT, 0);
  }
  if (st.pan_src) {
    for (int q = pwave; q < MB; q += MEGA_THREADS / 64)
      dma_row<COH>(st.pan_src + (long)q * st.pan_ld + 2 * lane, S + pa_off + q * LDP);
  }
  if (wave >= 4 && wave <= 6) {
    for (int d = 0; d < npre; ++d) {
      dma_slab(slab_begin + d);
      if (slab_begin + d < s_fresh) dma_cslab(slab_begin + d);
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  const long long t_spin0 = wall_clock64();

  if (wave == 7) {
    // ---- the poller: raises c_stop_req when the workgroup's next chain task has become ready
    if (has_intr) {
      for (;;) {
        const int fin = rfl(lds_ld(ctl + C_FIN) | lds_ld(ctl + C_ABORT));
        if (fin) break;
        int ok = 0;
        if (lane == 0) ok = cond_ok(ic) ? 1 : 0;
        ok = rfl(ok);
        if (ok) { if (lane == 0) lds_st(ctl + C_STOPREQ, 1); break; }
        __builtin_amdgcn_s_sleep(8);
      }
    }
  } else if (wave >= 4) {
    // ---- the movers
    const bool lead = wave == 4;
    int pub = npre;          // slabs of this wave published in c_land
    int issued = npre;       // slabs of this wave issued
    bool stopped = false;
    for (int t = slab_begin + npre; t < nslabs; ++t) {
      if (lead) {
        const int need = t - nbuf - slab_begin + 1;   // every consumer has read the slab whose buffer slab t reuses
        int it = 0;
        for (;;) {
          cfence();
          const int r0 = lds_ld(ctl + C_READ0), r1 = lds_ld(ctl + C_READ0 + 1), r2 = lds_ld(ctl + C_READ0 + 2), r3 = lds_ld(ctl + C_READ0 + 3);
          const int rq = rfl(lds_ld(ctl + C_STOPREQ) | lds_ld(ctl + C_ABORT));
          const int rd = rfl(min(min(r0, r1), min(r2, r3)));
          cfence();
          if (rq) { stopped = true; break; }
          if (rd >= need) break;
          __builtin_amdgcn_s_sleep(1);
          if ((++it & 1023) == 0 && wall_clock64() - t_spin0 > SPIN_TICKS) { if (lane == 0) lds_st(ctl + C_ABORT, 1); stopped = true; break; }
        }
        if (stopped) {
          if (lane == 0) lds_st(ctl + C_STOPAT, t);
          cfence();
          if (lane == 0) lds_st(ctl + C_FIN, 1);
          break;
        }
        if (lane == 0) lds_st(ctl + C_ISSUE, t + 1);
      } else {
        int it = 0;
        for (;;) {
          cfence();
          const int is = rfl(lds_ld(ctl + C_ISSUE)), fin = rfl(lds_ld(ctl + C_FIN) | lds_ld(ctl + C_ABORT));
          cfence();
          if (is > t) break;
          if (fin) { stopped = true; break; }
          __builtin_amdgcn_s_sleep(1);
          if ((++it & 1023) == 0 && wall_clock64() - t_spin0 > SPIN_TICKS) { if (lane == 0) lds_st(ctl + C_ABORT, 1); stopped = true; break; }
        }
        if (stopped) break;
      }
      dma_slab(t);
      // everything older than the instructions just issued has landed (LDS-DMA loads complete in order)
      if (t < s_fresh) {
        dma_cslab(t);
        if (wave == 6) __builtin_amdgcn_s_waitcnt(0x0F7C); else __builtin_amdgcn_s_waitcnt(0x0F7E);   // vmcnt(12) / vmcnt(14)
      } else {
        if (wave == 6) __builtin_amdgcn_s_waitcnt(0x0F7A); else __builtin_amdgcn_s_waitcnt(0x0F7B);   // vmcnt(10) / vmcnt(11)
      }
      ++issued;
      cfence();
      if (issued - 1 > pub) { pub = issued - 1; lds_st_raw(ctl + C_LAND0 + (wave - 4), pub); }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the last slab(s) of this wave
    cfence();
    if (lane == 0 && issued > pub) lds_st(ctl + C_LAND0 + (wave - 4), issued);
    if (lead && !stopped && lane == 0) lds_st(ctl + C_FIN, 1);
  } else {
    // ---- the consumers
    double fa[NBK / 4];
    {
      const double* pa = S + pa_off + (16 * mt + c) * LDP + g;
#pragma unroll
      for (int kk = 0; kk < NBK / 4; ++kk) fa[kk] = pa[4 * kk];
    }
    d4 sq = sq_io;
    long long acc_wait = 0, acc_all = 0, nslab_run = 0, nslow = 0;
    gdouble* cp[4];
    gdouble* cp2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long off = (long)(16 * mt + g + 4 * r) * st.ldc + n0 + (long)slab_begin * NS + 16 * nt + c;
      cp[r] = Cg + off;
      cp2[r] = Cg2 ? Cg2 + off : nullptr;
    }
    const int pb_lane = (16 * nt + c) * LDP + g;
    const int pc_lane = (16 * mt + g) * NS + 16 * nt + c;   // this lane's element of row r: + 4 r NS
    double pend[4] = {0.0, 0.0, 0.0, 0.0};
    auto finish_prev = [&](const d4& P0, const d4& P1, const double* cold, int ps) {   // ps: the slab these sums belong to
      const bool has_old = MODE == MODE_SUB || ps < s_fresh;   // (wave-uniform)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double o = P0[r] + P1[r];
        const double cv = has_old ? cold[r] : 0.0;
        pend[r] = MODE == MODE_FIN ? o : (MODE == MODE_SUB ? cv - o : cv + o);
      }
    };
    auto store_r = [&](int r, int ps) {   // ps: the slab the pending values belong to
      if constexpr (MODE == MODE_SQ) {
        sq[r] += pend[r] * pend[r];
      } else {
        if constexpr (MODE == MODE_FIN) S[OFF_PB + (16 * mt + g + 4 * r) * LDP + ps * NS + 16 * nt + c] = pend[r];
        if (!(dbg & 2)) st_c<COH>(cp[r], pend[r]);
        if constexpr (MODE == MODE_FIN) { if (Cg2) st_c<COH>(cp2[r], pend[r]); }
      }
      cp[r] += NS;
      if constexpr (MODE == MODE_FIN) cp2[r] += NS;
    };
    double fb[2][8];
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    // Blocks until slab t has landed; false when the stream ends before it (stop / abort / bounded spin ran out).
    auto wait_landed = [&](int t) -> bool {
      const int need = t - slab_begin + 1;
      int it = 0;
      bool ok = true;
      long long t_w0 = 0;
      if (kGpkExp && st.tacc) t_w0 = __builtin_readcyclecounter();
      for (;;) {
        cfence();
        const int l0 = lds_ld(ctl + C_LAND0), l1 = lds_ld(ctl + C_LAND0 + 1), l2 = lds_ld(ctl + C_LAND0 + 2);
        const int sa = rfl(lds_ld(ctl + C_STOPAT)), ab = rfl(lds_ld(ctl + C_ABORT));
        const int ln = rfl(min(l0, min(l1, l2)));
        cfence();
        if ((t >= sa) | ab) { ok = false; break; }
        if (ln >= need) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 1023) == 0) {
          const bool late = wall_clock64() - t_spin0 > SPIN_TICKS;
          // (the clock read is a scalar-memory access; they complete out of order, so with one possibly in flight every later
          // LDS wait would have to be a full one -- settle it here, on the slow path)
          __builtin_amdgcn_s_waitcnt(0xC07F);
          if (late) { if (lane == 0) lds_st(ctl + C_ABORT, 1); ok = false; break; }
        }
      }
      if (kGpkExp && st.tacc) { acc_wait += __builtin_readcyclecounter() - t_w0; nslow += 1; __builtin_amdgcn_s_waitcnt(0xC07F); }
      return ok;
    };
    // One slab.  CUR accumulates it; PRV still holds the sums of slab s - 1, which are folded into C under the first MFMAs of this
    // slab (no MFMA result is ever waited for inside the loop); coldc / coldp: the old C values of this / the previous slab.
    // fb[0] holds this slab's first fragments on entry (requested by the previous slab; the first slab of a call requests its
    // own).  Returns 0 when the next slab's first fragments are on their way, -1 when the stream ends with this slab.
    auto body = [&](auto first_tag, int s_in, d4& cur0, d4& cur1, const d4& prv0, const d4& prv1, double* coldc, const double* coldp) -> int {
      constexpr bool have_prev = !decltype(first_tag)::value;   // the first slab of a call is peeled: nothing to fold yet
      const int s = rfl(s_in);
      long long t_top = 0;
      if (kGpkExp && st.tacc) t_top = __builtin_readcyclecounter();
      const double* pb = S + buf_off(s) + pb_lane;
      if constexpr (!have_prev) {
#pragma unroll
        for (int k = 0; k < 8; ++k) fb[0][k] = pb[4 * k];
      }
      // The compiler waits for ALL outstanding LDS traffic at the first use of any LDS result (it does not count past a loop
      // header), so every group issues its first MFMA pair BEFORE the LDS requests of the next group: at the next group's first
      // MFMA the youngest request is then three MFMA pairs (~400 cycles) old and the full wait falls through.
#define MFMA2(K, FB)                                                                     \
  cur0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[(K)], FB[(K) & 7], cur0, 0, 0, 0);      \
  cur1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[(K) + 1], FB[((K) & 7) + 1], cur1, 0, 0, 0)
      // ---- group 0 (k = 0 .. 7): fragments of group 1 requested, the previous slab folded into C in the MFMAs' shadow
      SB();
      cur0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0], fb[0][0], zero, 0, 0, 0);
      cur1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1], fb[0][1], zero, 0, 0, 0);
      SB();
#pragma unroll
      for (int k = 0; k < 8; ++k) fb[1][k] = pb[4 * (8 + k)];
      if constexpr (have_prev) finish_prev(prv0, prv1, coldp, s - 1);
      SB();
      MFMA2(2, fb[0]);
      SB();
      if constexpr (have_prev) { store_r(0, s - 1); store_r(1, s - 1); }
      SB();
      MFMA2(4, fb[0]);
      SB();
      if constexpr (have_prev) { store_r(2, s - 1); store_r(3, s - 1); }
      SB();
      MFMA2(6, fb[0]);
      SB();
      // ---- group 1 (k = 8 .. 15): fragments of group 2 and this slab's old C values requested
      MFMA2(8, fb[1]);
      SB();
#pragma unroll
      for (int k = 0; k < 8; ++k) fb[0][k] = pb[4 * (16 + k)];
      if constexpr (MODE != MODE_FIN) {   // (unconditional: a slab without old values reads stale LDS and drops it in finish_prev)
        const double* pc = S + cbuf_off(s) + pc_lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) coldc[r] = pc[4 * r * NS];
      }
      SB();
      MFMA2(10, fb[1]);
      MFMA2(12, fb[1]);
      MFMA2(14, fb[1]);
      SB();
      // ---- group 2 (k = 16 .. 23): the last fragments of the slab requested -> its buffers may be refilled; the next slab's
      // state is read here and looked at one group later
      MFMA2(16, fb[0]);
      SB();
#pragma unroll
      for (int k = 0; k < 8; ++k) fb[1][k] = pb[4 * (24 + k)];
      cfence();
      if (lane == 0) lds_st(ctl + C_READ0 + wave, s - slab_begin + 1);
      const int q0 = lds_ld(ctl + C_LAND0), q1 = lds_ld(ctl + C_LAND0 + 1), q2 = lds_ld(ctl + C_LAND0 + 2);
      const int qs = lds_ld(ctl + C_STOPAT), qa = lds_ld(ctl + C_ABORT);
      cfence();
      SB();
      MFMA2(18, fb[0]);
      MFMA2(20, fb[0]);
      MFMA2(22, fb[0]);
      SB();
      // ---- group 3 (k = 24 .. 31): the next slab's first fragments requested (in the steady state it landed long ago)
      MFMA2(24, fb[1]);
      SB();
      int nx = -1;
      if (s + 1 < nslabs) {
        const bool fast = (rfl(min(q0, min(q1, q2))) >= s + 2 - slab_begin) & (s + 1 < rfl(qs)) & (rfl(qa) == 0);
        if (fast || wait_landed(s + 1)) nx = 0;
      }
      {
        // (requested unconditionally -- when there is no next slab the values are never used)
        const double* pbn = S + buf_off(s + 1) + pb_lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) fb[0][k] = pbn[4 * k];
      }
      SB();
      MFMA2(26, fb[1]);
      MFMA2(28, fb[1]);
      MFMA2(30, fb[1]);
      SB();
#undef MFMA2
      if (kGpkExp && st.tacc) { acc_all += __builtin_readcyclecounter() - t_top; nslab_run += 1; }
      return nx;
    };
    d4 a0 = zero, a1 = zero, b0 = zero, b1 = zero;
    double ca[4] = {0.0, 0.0, 0.0, 0.0}, cb[4] = {0.0, 0.0, 0.0, 0.0};
    int s = slab_begin;
    if (wait_landed(s)) {
      // (slab k = s - slab_begin: accumulators / old-C registers a for even k, b for odd k)
      int h = rfl(body(TagTrue{}, s, a0, a1, b0, b1, ca, cb));
      ++s;
      while (h >= 0) {
        h = rfl(body(TagFalse{}, s, b0, b1, a0, a1, cb, ca)); ++s; if (h < 0) break;
        h = rfl(body(TagFalse{}, s, a0, a1, b0, b1, ca, cb)); ++s;
      }
    }
    if (s > slab_begin) {   // the last slab run (s - 1) is still in its accumulators
      const int k = s - 1 - slab_begin;
      if ((k & 1) == 0) finish_prev(a0, a1, ca, s - 1); else finish_prev(b0, b1, cb, s - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) store_r(r, s - 1);
    }
    if (kGpkExp && st.tacc && tid == 0) { st.tacc[0] += acc_wait; st.tacc[1] += acc_all; st.tacc[2] += nslow; st.tacc[4] += nslab_run; }
    sq_io = sq;
  }
  __syncthreads();
  const int sa = rfl(lds_ld(ctl + C_STOPAT)), ab = rfl(lds_ld(ctl + C_ABORT));
  if (ab) return -1;
  return sa < nslabs ? sa : nslabs;
}

// ---- chain task bookkeeping -------------------------------------------------------------------------------------------
// Tasks of panel p, in dependency order:  LEAF(p);  then for block rows i = p+1 .. nb-1 the four quarter tasks FIN(i, u, p)
// (rows 128 i + 32 u .. + 31:  X[r, p] <- X[r, p] inv(L_pp)^T) and UPD(i, u, p) (X[r, n] -= X[r, p] L[n, p]^T for the
// columns n after panel p up to the rows' own diagonal).  Two classes, two pools of workgroups:
//   NEAR  LEAF(p) and the block rows p+1 .. p+3: short tasks (<= 12 slabs) on or next to the critical path
//         leaf -> FIN(p+1) -> UPD(p+1) -> leaf; dealt to the first GA workgroups;
//   FAR   block rows p+4 ..: long tasks (up to 60 slabs) that have several panels of slack; dealt to the other workgroups.
// A workgroup runs its tasks in the global (panel, slot) order.  Keeping the classes apart keeps a near task from waiting
// behind a far one of an earlier panel in the same workgroup's list (first version: one pool, the factorisation of
// n = 2048 took 170 us per panel instead of ~55).
enum { T_LEAF = 0, T_FIN = 1, T_UPD = 2 };
struct Task { int type, p, i, u, c; };   // c: column chunk of a far UPD
constexpr int NEAR_ROWS = 3;
constexpr int CH = 16;   // slabs per far UPD task: a far update is split along its columns into independent chunks, so that
                         // the chain FIN(i,u,p) -> UPD(i,u,p) -> FIN(i,u,p+1) of one row quarter costs a chunk, not up to 60
                         // slabs, per panel (unsplit, block row 15 alone needed 2.4 ms for a 2048 factorisation)

__device__ __forceinline__ int near_rows(int nb, int p) { const int r = nb - 1 - p; return r < NEAR_ROWS ? r : NEAR_ROWS; }
__device__ __forceinline__ int ntasks_near(int nb, int p) { return 1 + 8 * near_rows(nb, p); }
// number of UPD tasks of row quarter (i, .) at panel p: 1 for a near block row, ceil(4 (i - p) / CH) chunks for a far one
__device__ __forceinline__ int upd_chunks(int i, int p) { return (i - p <= NEAR_ROWS) ? 1 : (4 * (i - p) + CH - 1) / CH; }
// completed UPD tasks of row quarter (i, .) once panels 0 .. p are applied (the threshold its next FIN / its LEAF waits for)
__device__ __forceinline__ int upd_cum(int i, int p) {
  int c = 0;
  for (int pp = 0; pp <= p; ++pp) c += upd_chunks(i, pp);
  return c;
}
__device__ __forceinline__ int ntasks_far(int nb, int p) {
  int n = 0;
  for (int i = p + 1 + NEAR_ROWS; i < nb; ++i) n += 4 + 4 * upd_chunks(i, p);
  return n;
}
__device__ __forceinline__ int slot_owner_offset(int p) { return 29 * p; }
__device__ __forceinline__ Task decode_near(int p, int t) {
  Task k{T_LEAF, p, p, 0, 0};
  if (t == 0) return k;
  const int rem = t - 1;                 // block row p + 1 + rem / 8: FIN x 4 then UPD x 4
  k.i = p + 1 + rem / 8;
  k.type = (rem % 8) < 4 ? T_FIN : T_UPD;
  k.u = rem % 4;
  return k;
}
__device__ __forceinline__ Task decode_far(int nb, int p, int t) {
  const int nrest = nb - 1 - p - NEAR_ROWS;   // block rows p + 4 ..: all FIN first, then the UPD chunks row by row
  Task k{T_FIN, p, 0, 0, 0};
  if (t < 4 * nrest) { k.i = p + 1 + NEAR_ROWS + t / 4; k.u = t % 4; return k; }
  t -= 4 * nrest;
  k.type = T_UPD;
  for (int i = p + 1 + NEAR_ROWS; i < nb; ++i) {
    const int n = 4 * upd_chunks(i, p);
    if (t < n) { k.i = i; k.c = t / 4; k.u = t % 4; return k; }
    t -= n;
  }
  return k;
}
// flags: leaf[p] = 1 once LEAF(p) is published; finc[p][i] = number of published FIN tasks of panel p in block rows
// p+1 .. i (every FIN(i', ., p) adds one to finc[p][i] for all i >= i'); rowd[i][u] = published UPD tasks of row quarter (i, u)
__device__ __forceinline__ void task_cond(const MegaArgs& a, const Task& k, Cond& c) {
  const int nb = a.nb;
  cond_init(c, a.flags);
  if (k.type == T_LEAF) {
    if (k.p > 0) {
      c.a0 = a.flags + f_rowd(nb, k.p, 0); c.a1 = a.flags + f_rowd(nb, k.p, 1);
      c.a2 = a.flags + f_rowd(nb, k.p, 2); c.a3 = a.flags + f_rowd(nb, k.p, 3);
      c.t0 = c.t1 = c.t2 = c.t3 = upd_cum(k.p, k.p - 1);
    }
  } else if (k.type == T_FIN) {
    c.a0 = a.flags + f_leaf(k.p); c.t0 = 1;
    c.a1 = a.flags + f_rowd(nb, k.i, k.u); c.t1 = k.p > 0 ? upd_cum(k.i, k.p - 1) : 0;
  } else {
    c.a0 = a.flags + f_finc(nb, k.p, k.i); c.t0 = 4 * (k.i - k.p);   // the B rows it reads: block rows p+1 .. i
  }
}

// (the leaf and the stream are real calls: each gets the whole register file instead of sharing it with the scheduler's state)
template <bool WT>
__device__ __noinline__ void mega_leaf(double* A, long lda, double* inv, int* info, int col0) {
  extern __shared__ __attribute__((aligned(16))) double S[];
  gpk_leaf::leaf_body<false, WT>(S, A, lda, NBK, inv, info, col0, nullptr);
}

template <int PROTO>
__device__ __forceinline__ void acquire_all() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// end of a task: every store of the workgroup is performed device-wide before thread 0 raises the flag
template <int PROTO>
__device__ __forceinline__ void publish_barrier() {
  __builtin_amdgcn_s_waitcnt(0);
  if constexpr (PROTO == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
}

template <int PROTO>
__global__ __launch_bounds__(MEGA_THREADS) void svgp_step_kernel(MegaArgs a) {
  extern __shared__ __attribute__((aligned(16))) double S[];
  __shared__ __attribute__((aligned(16))) int ctl[16];
  constexpr bool WT = PROTO == 1;   // chain data: write-through stores, agent-scope loads
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x, G = gridDim.x;
  const int nb = a.nb, m = a.m, P = a.P;
  const long ld = a.ld;
  double* E = a.T + (long)m * ld;
  const bool has_rows = wg < a.nbulk;
  const int r0 = wg * MB;
  const int nr = has_rows ? (a.rows - r0 < MB ? a.rows - r0 : MB) : 0;
  double* E0 = E + (long)r0 * ld;
  int* abortf = a.flags + f_abort(nb);

  int q = 0, ph = 0, pos = 0;   // bulk: panel, phase (0 FIN, 1..P PROJ of latent ph-1, P+1 UPD), next slab
  bool panel_ok = false;        // the A panel of the bulk row block (A^T[:, q]) is resident in LDS (OFF_PB)
  int cp = 0, ct = -1;          // chain: panel cursor, slot cursor within the panel (-1: not yet computed)
  // near pool: workgroups [0, GA); far pool: [GA, G) (one pool for tiny grids: then everybody takes near AND far tasks,
  // far ones as extra slots behind the near ones -- handled by giving such grids the near decode over both ranges)
  const int GA = (G >= 16 && !a.one_pool) ? (G / 4 > 32 ? G / 4 : 32 < G / 2 ? 32 : G / 2) : G;
  const bool two_pools = GA < G;
  const bool is_near = !two_pools || wg < GA;
  const int pool_n = is_near ? GA : G - GA;
  const int pool_id = is_near ? wg : wg - GA;
  d4 sq = {0.0, 0.0, 0.0, 0.0};
  const long long t_start = wall_clock64();
  bool aborted = false;
  // Roles.  The four waves that issue the MFMAs of a slab must sit on four DIFFERENT SIMDs (a SIMD has one matrix pipe): the
  // hardware places the eight waves of a workgroup two per SIMD but in no documented order, so every wave reads its SIMD id
  // and the lower-numbered wave of each SIMD becomes a consumer (role 0..3), the other one a producer (role 4..7).  (With
  // the roles fixed as waves 0..3 / 4..7 a slab took 2.2 us against 0.9 us of MFMA issue time: pairs of consumers shared a pipe.)
  int vw = wave;
  {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (lane == 0) ctl[wave] = (int)((hw >> 4) & 3u);
    __syncthreads();
    int simd[8], cons[8], ncons = 0;
    for (int w2 = 0; w2 < 8; ++w2) simd[w2] = ctl[w2];
    for (int w2 = 0; w2 < 8; ++w2) {
      cons[w2] = 1;
      for (int w3 = 0; w3 < w2; ++w3) cons[w2] &= (simd[w3] != simd[w2]);
      ncons += cons[w2];
    }
    if (ncons == 4 && a.role_map) {
      int rank = 0;
      for (int w2 = 0; w2 < wave; ++w2) rank += (cons[w2] == cons[wave]);
      vw = cons[wave] ? rank : 4 + rank;
    }
    __syncthreads();
  }
#ifdef GPK_EXPERIMENTAL
  if (a.trace && wg == a.trace_wg && lane == 0) {
    const long long idx = __hip_atomic_fetch_add((unsigned long long*)a.trace, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (idx < 4096) {
      long long* r = a.trace + 1 + 8 * idx;
      unsigned hw2;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw2));
      r[0] = wall_clock64(); r[1] = r[0]; r[2] = wg; r[3] = 9; r[4] = wave; r[5] = (hw2 >> 4) & 3; r[6] = vw; r[7] = hw2;
    }
  }
#endif
  // whatever an earlier kernel left in this XCD's L2 / this CU's L1 of the buffers that are rewritten below is dropped once,
  // here; after that, protocol 1 never invalidates a cache on behalf of the minibatch rows again
  acquire_all<PROTO>();

  for (;;) {
    // ---- my next chain task ---------------------------------------------------------------------------------------
    bool have_task = false;
    Task tk{};
    while (cp < nb) {
      const int nn = ntasks_near(nb, cp), nf = ntasks_far(nb, cp);
      const int nt_p = two_pools ? (is_near ? nn : nf) : nn + nf;   // one pool: the far slots follow the near ones
      if (ct < 0) ct = ((pool_id - slot_owner_offset(cp)) % pool_n + pool_n) % pool_n;
      if (ct < nt_p) {
        if (two_pools) tk = is_near ? decode_near(cp, ct) : decode_far(nb, cp, ct);
        else tk = ct < nn ? decode_near(cp, ct) : decode_far(nb, cp, ct - nn);
        have_task = true;
        break;
      }
      ++cp; ct = -1;
    }
    const bool bulk_left = has_rows && q < nb;
    if (!have_task && !bulk_left) break;
    Cond tc;
    cond_init(tc, a.flags);
    if (have_task) task_cond(a, tk, tc);
    // ---- decide: 1 chain task, 2 bulk quantum, 0 nothing ready, 3 abort -------------------------------------------
    if (tid == 0) {
      int d = 0;
      if (ld_flag(abortf)) d = 3;
      else if (have_task && cond_ok(tc)) d = 1;
      else if (bulk_left) {
        bool ok = true;
        if (ph == 0) ok = ld_flag(a.flags + f_leaf(q)) >= 1;
        else if (ph == P + 1 && q < nb - 1) ok = ld_flag(a.flags + f_finc(nb, q, nb - 1)) >= 4 * (nb - q - 1);
        if (ok) d = 2;
      }
      if (d == 0 && wall_clock64() - t_start > a.timeout_ticks) { st_flag(abortf, 1); d = 3; }
      ctl[0] = d;
    }
    __syncthreads();
    const int d = ctl[0];
    __syncthreads();
    if (d == 3) { aborted = true; break; }
    if (d == 0) { __builtin_amdgcn_s_sleep(8); continue; }

    // One stream per loop iteration, described by the branch that owns it and run at a single call site per coherence
    // variant (the leaf and the stream body are large: one inlined copy each keeps the kernel's register budget for them).
    const int p = tk.p;
    const long long t_task0 = wall_clock64();
    const long long c_task0 = (long long)__builtin_readcyclecounter();
    Stream st{};
    st.nrows = MB; st.pa_off = OFF_PA; st.nfresh = 0;
    st.dbg = (d == 2) ? a.dbg : 0;
    st.tacc = (kGpkExp && a.trace && d == 2 && wg == a.trace_wg) ? a.trace + 1 + 8 * 4096 : nullptr;   // (what-if switches act on the minibatch rows only: the factorisation stays exact)
    bool coh = false, run = true;
    int begin = 0;
    const Cond* intr = nullptr;
    int what;   // 0 leaf, 1 chain FIN, 2 chain UPD, 3 bulk FIN, 4 bulk PROJ, 5 bulk UPD, 6 bulk final
    if (d == 1) {
      // ================= chain task =================
      if (tk.type == T_LEAF) {
        what = 0; run = false;
        if (tid == 0) a.stamps[2 * p] = wall_clock64();
        acquire_all<PROTO>();
        mega_leaf<WT>(a.T + (long)(NBK * p) * (ld + 1), ld, a.invd + (long)p * NBK * NBK, a.info, NBK * p);
      } else {
        double* R = a.T + (long)(NBK * tk.i + MB * tk.u) * ld;   // the task's 32 rows of the square part
        coh = WT;
        if constexpr (!WT) acquire_all<PROTO>();
        st.pan_src = R + NBK * p; st.pan_ld = ld;   // the task's A panel: its rows of column block p
        if (tk.type == T_FIN) {
          what = 1;
          st.B = a.invd + (long)p * NBK * NBK; st.ldb = NBK; st.n0 = 0; st.n1 = NBK;
          st.C = R + NBK * p; st.ldc = ld; st.mode = MODE_FIN;
          st.C2 = a.Lfin + (long)(NBK * tk.i + MB * tk.u) * ld + NBK * p;
        } else {
          what = 2;
          const int n_end = NBK * tk.i + MB * tk.u + MB;   // the rows' own diagonal
          st.B = a.T + NBK * p; st.ldb = ld; st.n0 = NBK * (p + 1); st.n1 = n_end;
          if (tk.i - p > NEAR_ROWS) {                     // far: column chunk tk.c of CH slabs (may be empty for small u)
            st.n0 += NS * CH * tk.c;
            st.n1 = st.n0 + NS * CH < n_end ? st.n0 + NS * CH : n_end;
            if (st.n1 <= st.n0) { run = false; st.pan_src = nullptr; }
          }
          st.C = R; st.ldc = ld; st.mode = MODE_SUB;
        }
      }
    } else {
      // ================= bulk quantum (d == 2) =================
      st.nrows = nr;
      if (ph == 0) {
        // FIN: A^T[:, q] = E[:, q] inv(L_qq)^T
        what = 3;
        if constexpr (!WT) acquire_all<PROTO>();   // (protocol 1: inv(L_qq) is write-once data, see "Coherence")
        st.pan_src = E0 + NBK * q; st.pan_ld = ld;
        st.B = a.invd + (long)q * NBK * NBK; st.ldb = NBK; st.n0 = 0; st.n1 = NBK;
        st.C = E0 + NBK * q; st.ldc = ld; st.mode = MODE_FIN;
      } else {
        st.pa_off = OFF_PB;
        if (!panel_ok) {   // A^T[:, q] again (own rows, written by this workgroup), fetched in the stream's prologue
          st.pan_src = E0 + NBK * q; st.pan_ld = ld;
          panel_ok = true;
        }
        begin = pos;
        if (ph <= P) {
          // PROJ, latent pl: C_pl[:, 0 : 128 (q+1)] += A^T[:, q] Lq_pl[q, :]; the last panel squares instead of storing
          what = 4;
          const int pl = ph - 1;
          const bool last = q == nb - 1;
          st.B = a.LqT + (long)pl * m * a.ldl + NBK * q; st.ldb = a.ldl; st.n0 = 0; st.n1 = NBK * (q + 1);
          st.C = a.Cacc + ((long)pl * a.rows_pad + r0) * ld; st.ldc = ld;
          st.mode = last ? MODE_SQ : MODE_ADD; st.nfresh = NBK * q;
          intr = (have_task && !last) ? &tc : nullptr;
        } else if (q < nb - 1) {
          // UPD: E[:, n] -= A^T[:, q] L[n, q]^T for the columns still to come
          what = 5;
          if constexpr (!WT) { if (pos == 0) acquire_all<PROTO>(); }
          st.B = (WT ? a.Lfin : a.T) + NBK * q; st.ldb = ld; st.n0 = NBK * (q + 1); st.n1 = m;
          st.C = E0; st.ldc = ld; st.mode = MODE_SUB;
          intr = have_task ? &tc : nullptr;
        } else {
          what = 6; run = false;
        }
      }
    }
    int endpos = 0;
    const int nslabs = run ? (st.n1 - st.n0 + NS - 1) / NS : 0;
    if (run) {
      // (one instantiation per coherence variant and epilogue: the mode is a compile-time constant inside the slab loop)
      if (a.sync_mode == 0) {   // (A/B: the barrier-per-slab version)
        if (coh) {
          if (st.mode == MODE_FIN) endpos = stream_run<true, MODE_FIN>(st, begin, intr, ctl, sq, vw);
          else endpos = stream_run<true, MODE_SUB>(st, begin, intr, ctl, sq, vw);
        } else if (st.mode == MODE_FIN) endpos = stream_run<false, MODE_FIN>(st, begin, intr, ctl, sq, vw);
        else if (st.mode == MODE_SUB) endpos = stream_run<false, MODE_SUB>(st, begin, intr, ctl, sq, vw);
        else if (st.mode == MODE_ADD) endpos = stream_run<false, MODE_ADD>(st, begin, intr, ctl, sq, vw);
        else endpos = stream_run<false, MODE_SQ>(st, begin, intr, ctl, sq, vw);
      } else {
        if (coh) {
          if (st.mode == MODE_FIN) endpos = stream_run_flags<true, MODE_FIN>(st, begin, intr, ctl, sq, vw);
          else endpos = stream_run_flags<true, MODE_SUB>(st, begin, intr, ctl, sq, vw);
        } else if (st.mode == MODE_FIN) endpos = stream_run_flags<false, MODE_FIN>(st, begin, intr, ctl, sq, vw);
        else if (st.mode == MODE_SUB) endpos = stream_run_flags<false, MODE_SUB>(st, begin, intr, ctl, sq, vw);
        else if (st.mode == MODE_ADD) endpos = stream_run_flags<false, MODE_ADD>(st, begin, intr, ctl, sq, vw);
        else endpos = stream_run_flags<false, MODE_SQ>(st, begin, intr, ctl, sq, vw);
      }
      if (endpos < 0) {   // a bounded spin inside the stream ran out: give up device-wide
        if (tid == 0) st_flag(abortf, 1);
        aborted = true;
        break;
      }
    }
    // ---- what follows the stream ----------------------------------------------------------------------------------
    if (what <= 2) {
      publish_barrier<PROTO>();
      if (tid == 0) {
        if (what == 0) { st_flag(a.flags + f_leaf(p), 1); a.stamps[2 * p + 1] = wall_clock64(); }
        else if (what == 2) add_flag(a.flags + f_rowd(nb, tk.i, tk.u), 1);
        // (what == 1, FIN: the cumulative counters of the block rows i .. nb-1, one lane each -- below)
      }
      if (what == 1 && tid < nb - tk.i) add_flag(a.flags + f_finc(nb, p, tk.i + tid), 1);
#ifdef GPK_EXPERIMENTAL
      if (a.trace && tid == 0) {
        const long long idx = __hip_atomic_fetch_add((unsigned long long*)a.trace, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (idx < 4096) {
          long long* r = a.trace + 1 + 8 * idx;
          r[0] = t_task0; r[1] = wall_clock64(); r[2] = wg; r[3] = what; r[4] = p; r[5] = tk.i; r[6] = tk.u;
          r[7] = (what == 2) ? ld_flag(a.flags + f_finc(nb, p, tk.i)) : (what == 1 ? ld_flag(a.flags + f_rowd(nb, tk.i, tk.u)) : 0);
        }
      }
#endif
      panel_ok = false;
      ct += pool_n;
      if (ct >= (two_pools ? (is_near ? ntasks_near(nb, cp) : ntasks_far(nb, cp)) : ntasks_near(nb, cp) + ntasks_far(nb, cp))) { ++cp; ct = -1; }
    }
#ifdef GPK_EXPERIMENTAL
    if (what >= 3 && a.trace && tid == 0 && wg == a.trace_wg) {
      const long long idx = __hip_atomic_fetch_add((unsigned long long*)a.trace, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (idx < 4096) {
        long long* r = a.trace + 1 + 8 * idx;
        r[0] = t_task0; r[1] = wall_clock64(); r[2] = wg; r[3] = what; r[4] = q; r[5] = begin; r[6] = endpos;
        r[7] = (long long)__builtin_readcyclecounter() - c_task0;   // shader-clock cycles of the quantum: cycles / us = clock in MHz
      }
    }
#endif
    if (what <= 2) {
    } else if (what == 3) {
      // s0[row] (+)= sum_k A^2, fmean[row][p] (+)= sum_k A[row][k] q_mu[128 q + k][p]: 16 threads per row
      const int row = tid >> 4, sub = tid & 15;
      const double* ap = S + OFF_PB + row * LDP + sub * 8;
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) s2 += ap[k] * ap[k];
      s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4); s2 += __shfl_xor(s2, 8);
      if (sub == 0 && row < nr) a.s0[r0 + row] = (q == 0 ? 0.0 : a.s0[r0 + row]) + s2;
      for (int pp = 0; pp < P; ++pp) {
        double mv = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) mv += ap[k] * a.q_mu[(long)(NBK * q + sub * 8 + k) * P + pp];
        mv += __shfl_xor(mv, 1); mv += __shfl_xor(mv, 2); mv += __shfl_xor(mv, 4); mv += __shfl_xor(mv, 8);
        if (sub == 0 && row < nr) {
          double* fm = a.fmean + (long)(r0 + row) * P + pp;
          *fm = (q == 0 ? 0.0 : *fm) + mv;
        }
      }
      panel_ok = true; ph = 1; pos = 0;
      sq = (d4){0.0, 0.0, 0.0, 0.0};
    } else if (what == 4) {
      pos = endpos;
      if (endpos >= nslabs) {
        if (q == nb - 1) {
          // ssq[pl][row]: the 16 column lanes of a row, then the two column halves (waves nt = 0, 1) in a fixed order
          double* red = S + OFF_MISC;
          if (vw < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              double v = sq[r];
              v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
              if ((lane & 15) == 0) red[((vw >> 1) & 1) * MB + 16 * (vw & 1) + (lane >> 4) + 4 * r] = v;
            }
          }
          __syncthreads();
          if (tid < nr) a.ssq[(long)(ph - 1) * a.rows + r0 + tid] = red[tid] + red[MB + tid];
          __syncthreads();
          sq = (d4){0.0, 0.0, 0.0, 0.0};
        }
        ++ph; pos = 0;
      }
    } else if (what == 5) {
      pos = endpos;
      if (endpos >= nslabs) { ++q; ph = 0; pos = 0; panel_ok = false; }
    } else {
      // ---- variational expectations of the row block (likelihoods/scalar_continuous.py:139-148), summed --------------
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();   // this workgroup's ssq / s0 / fmean stores are performed
      double* red = S + OFF_MISC;
      if (tid < MB) {
        double acc = 0.0;
        if (tid < nr) {
          const double c0 = -0.5 * 1.8378770664093453 - 0.5 * log(a.noise);
          const long b = r0 + tid;
          for (int pp = 0; pp < P; ++pp) {
            const double fv = a.variance - a.s0[b] + a.ssq[(long)pp * a.rows + b];
            const double dy = a.Y[b * a.ldy + pp] - (a.fmean[b * P + pp] + a.mean_const);
            acc += c0 - 0.5 * (dy * dy + fv) / a.noise;
          }
        }
        red[tid] = acc;
      }
      __syncthreads();
      if (tid == 0) {
        double tot = 0.0;
        for (int i = 0; i < MB; ++i) tot += red[i];
        __hip_atomic_store(a.partial + wg, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int prev = add_flag(a.flags + f_done(nb), 1);
        if (prev == a.nbulk - 1) {   // the last row block: the step's data term, summed in workgroup order
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          double o = 0.0;
          for (int i = 0; i < a.nbulk; ++i) o += __hip_atomic_load(a.partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          a.out[0] = o;
        }
      }
      __syncthreads();
      q = nb;
    }
  }
  if (aborted && tid == 0 && a.info) a.info[0] = INT_MAX;   // timed out: the caller sees a failed factorisation
}

}  // namespace

// flags + partials the step kernel needs (bytes), and the launch.  The caller has built Kuu (+ jitter) into T[0:m], Kfu into
// T[m:m+rows], LqT, and zeroes `flags` on the same stream before this launch.
size_t gpk_mega_flag_ints(int m) {
  const int nb = m / NBK;
  return (size_t)nb * nb + 6 * nb + 8 + 4 * (size_t)nb + 2;   // flag words, then (8-byte aligned) the leaf time stamps
}
int gpk_mega_supported(int m, int rows, int P, int ncu) {
  if (m < NBK || (m % NBK) || m / NBK > 64 || rows < 1 || P < 1 || P > 16) return 0;
  if ((rows + MB - 1) / MB > ncu) return 0;   // one row block per resident workgroup
  return 1;
}
#ifdef GPK_EXPERIMENTAL
namespace { long long* g_trace = nullptr; }
extern "C" __attribute__((visibility("default"))) int gpk_exp_mega_trace_dump(void) {
  if (!g_trace) return 0;
  static long long host[1 + 8 * 4096 + 8];
  if (hipMemcpy(host, g_trace, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  const long long n = host[0] < 4096 ? host[0] : 4096;
  long long t0 = 0;
  for (long long i = 0; i < n; ++i) if (i == 0 || host[1 + 8 * i] < t0) t0 = host[1 + 8 * i];
  {
    const long long* t = host + 1 + 8 * 4096;
    if (t[4] > 0)
      printf("# consumer loop of the traced workgroup, shader-clock cycles per slab over %lld slabs: before MFMA %.0f, MFMA section %.0f, "
             "epilogue %.0f, barrier wait %.0f   (flag-synchronised stream: waiting %.0f of %.0f cycles per slab, %lld blocking waits)\n", t[4],
             (double)t[0] / t[4], (double)t[1] / t[4], (double)t[2] / t[4], (double)t[3] / t[4], (double)t[0] / t[4], (double)t[1] / t[4], t[2]);
  }
  printf("# task trace: start_us dur_us wg what(0 leaf,1 FIN,2 UPD | 3 bulk FIN,4 PROJ,5 UPD,6 final: p=q i=first slab u=end slab flag=slabs)   (%lld records)\n", host[0]);
  for (long long i = 0; i < n; ++i) {
    const long long* r = host + 1 + 8 * i;
    printf("T %8.1f %6.1f wg=%3lld what=%lld p=%lld i=%lld u=%lld flag=%lld\n", (r[0] - t0) / 100.0, (r[1] - r[0]) / 100.0, r[2], r[3], r[4], r[5],
           r[6], r[7]);
  }
  fflush(stdout);
  return (int)n;
}
#endif

int gpk_launch_svgp_mega(hipStream_t s, int proto, int ncu, double* T, long ld, int m, int rows, double* invd, double* Lfin, const double* LqT,
                         long ldl, double* Cacc, const double* q_mu, int P, const double* Y, long ldy, double* s0, double* fmean,
                         double* ssq, double* partial, int* flags, int* info, double* out, double variance, double noise,
                         double mean_const, int min_wgs) {
  if (!gpk_mega_supported(m, rows, P, ncu)) return GPK_E_UNSUPPORTED;
  static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(svgp_step_kernel<0>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)MEGA_LDS);
  static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(svgp_step_kernel<1>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)MEGA_LDS);
  GPK_HIP(attr0);
  GPK_HIP(attr1);
  MegaArgs a{};
  a.T = T; a.ld = ld; a.invd = invd; a.Lfin = Lfin; a.LqT = LqT; a.ldl = ldl; a.Cacc = Cacc; a.q_mu = q_mu; a.Y = Y; a.ldy = ldy;
  a.s0 = s0; a.fmean = fmean; a.ssq = ssq; a.partial = partial; a.flags = flags; a.info = info; a.out = out;
  {
    const size_t nflag = (size_t)(m / NBK) * (m / NBK) + 6 * (m / NBK) + 8;
    a.stamps = (long long*)(flags + ((nflag + 1) & ~(size_t)1));
  }
  a.m = m; a.nb = m / NBK; a.rows = rows; a.P = P; a.nbulk = (rows + MB - 1) / MB;
  a.rows_pad = a.nbulk * MB;
  a.variance = variance; a.noise = noise; a.mean_const = mean_const;
  a.timeout_ticks = 200000000LL;   // 2 s of the 100 MHz wall clock
  a.one_pool = GPK_TUNE(MEGA_ONE_POOL, 0);
  a.role_map = GPK_TUNE(MEGA_ROLE_MAP, 1);
  a.sync_mode = GPK_TUNE(MEGA_SYNC, 0);   // (measured: the barrier version is the faster one, 3.10 against 3.28 ms at Cm)
  a.dbg = kGpkExp ? GPK_TUNE(MEGA_DBG, 0) : 0;
  int G = a.nbulk;
  if (G < min_wgs) G = min_wgs;
  if (G < 16) G = 16;   // two task pools need a few workgroups each
  if (G > ncu) G = ncu;
  if (G < 16) return GPK_E_UNSUPPORTED;
  GPK_HIP(hipMemsetAsync(flags, 0, gpk_mega_flag_ints(m) * sizeof(int), s));
#ifdef GPK_EXPERIMENTAL
  if (GPK_TUNE(MEGA_TRACE, 0)) {
    if (!g_trace) GPK_HIP(hipMalloc(&g_trace, sizeof(long long) * (1 + 8 * 4096 + 8)));
    GPK_HIP(hipMemsetAsync(g_trace, 0, sizeof(long long) * (1 + 8 * 4096 + 8), s));
    a.trace = g_trace;
    a.trace_wg = GPK_TUNE(MEGA_TRACE_WG, 100);
  }
#endif
  // (bench roofline leg: HIP events around the launch; algorithmic flops of everything the kernel does, M^3/3 + M^2 B (1 + P))
  const int prof = gpk_prof_begin(s, (double)m * m * m / 3.0 + (double)m * m * (double)rows * (1.0 + P), 7);
  if (proto == 1) hipLaunchKernelGGL((svgp_step_kernel<1>), dim3((unsigned)G), dim3(MEGA_THREADS), MEGA_LDS, s, a);
  else hipLaunchKernelGGL((svgp_step_kernel<0>), dim3((unsigned)G), dim3(MEGA_THREADS), MEGA_LDS, s, a);
  GPK_LAUNCH_CHECK();
  gpk_prof_end(prof, s);
  return 0;
}
