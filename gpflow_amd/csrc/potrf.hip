// Host-side orchestration (no device code here): the trapezoidal blocked Cholesky, triangular solves
// against a cached factor, the projection onto q_sqrt, and the two fused model drivers
// (GPR.log_marginal_likelihood, one shard of SVGP.elbo).
//
// Trapezoidal Cholesky.  A is [(n + extra) x n]: the top square block is factored, the `extra` rows
// below ride along through every panel solve and trailing update and come out as  B L^-T  -- the
// tf.linalg.triangular_solve of the reference fused into the factorisation.  Two-level right-looking:
//   outer panels of 640 columns (n >= 4096) -> trailing update is a K = 640 MFMA GEMM,
//   inner blocks of NB = 128 columns        -> leaf kernel (L11 and L11^-1), in-place panel solve
//                                              A21 <- A21 * L11^-T as a GEMM, update of the rest of the panel.
// For n < 4096 (the SVGP sizes) the outer panel IS one 128-column block: the factorisation is a latency chain
// leaf -> panel solve -> strip, and everything that is not on that chain (the solve of the minibatch rows) runs beside it
// as bulk work on a stream of its own.
#include "gpk_internal.h"
#include <algorithm>
#include <functional>
#include <mutex>
#include <vector>

namespace {
constexpr int NB = GPK_NB;
constexpr int NBO = 512;  // column group of the right-looking row solves (extra rows, gpk_trsm)

inline GemmArgs gemm_base(int m, int n, int k, double alpha, const double* A, long lda,
                          const double* B, long ldb, double beta, double* C, long ldc, int batch,
                          long sA, long sB, long sC) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.strideA = sA;
  g.B = B; g.ldb = ldb; g.strideB = sB;
  g.C = C; g.ldc = ldc; g.strideC = sC;
  g.m = m; g.n = n; g.k = k; g.alpha = alpha; g.beta = beta;
  g.b_tri_rows = n; g.batch = batch > 0 ? batch : 1;
  return g;
}

// how a bulk GEMM beside the latency chain is launched
struct Bulk {
  int cap = 0;    // cap on the persistent workgroups of the big (K >= 256) updates, 0 = one workgroup per tile
  int group_cap = 0;   // cap on the workgroups of the fused in-group solve, 0 = one per 16-row sliver
  int kmin = 256;      // updates with K below this are not capped
  int queue_cus = 0;   // > 0: the big updates run as persistent workgroups fed from a tile queue, two per compute unit of this many CUs
  void apply(GemmArgs& g) const {
    if (cap > 0 && g.k >= kmin) g.max_wgs = cap;
    if (queue_cus > 0 && g.k >= kmin) { g.tile_queue = 1; g.stagger_first = queue_cus; }
  }
};
}  // namespace

extern "C" const char* gpk_version(void) {
#ifdef GPK_EXPERIMENTAL
  return "gpk 0.4 (gfx950, fp64 MFMA) [A/B build: environment tunables enabled]";
#else
  return "gpk 0.4 (gfx950, fp64 MFMA)";
#endif
}

extern "C" size_t gpk_invd_elems(int n, int batch) {
  return (size_t)(batch > 0 ? batch : 1) * gpk_cdiv(n, NB) * NB * NB;
}

// ---- per-device internal state (created lazily, once; see gpk.h "Internal state and threading") -------------------
// The factorisation runs on streams of its own, forked from / joined to the caller's stream with events only:
//   P   "panel" stream, high priority: the latency-bound critical path (leaf, panel solve, inner updates, strip) of
//       the NEXT outer panel (look-ahead);
//   B   bulk stream, CU-masked in hardware: its mask leaves 8 compute units (one per XCD; mask bit i is CU i/8 of XCD
//       i%8 on MI355X, tools/cumask_test.hip) to the panel stream -- without that the one-workgroup leaf kernel, which
//       needs a whole CU's LDS, queues behind thousands of resident GEMM workgroups (a 2 ms stall per panel at
//       N = 16384) and the look-ahead never overlaps.  Large factorisations (n >= 4096) only: the big MFMA GEMMs of the
//       outer trailing updates and of the extra rows;
//   Bs  rest-updates of SMALL factorisations and of the single-leaf panels at the end of large ones (they are ON the
//       critical path there): all CUs.  (Rounds 1-2 ran that end of a large factorisation with wide panels and a second
//       masked stream over half the CUs; round 3 measured every hand-off from that stream to P at ~55 us while both are
//       busy -- their hardware queues share a microengine pipe -- and replaced it: potrf_core, "Panel boundaries".)
//   X   bulk stream of small factorisations: the right-looking solve of the extra rows (the SVGP minibatch).  Unmasked:
//       CU-masked queues dispatch its short kernels slowly and quantise its big updates badly (profiles/r03_*).
// One std::recursive_mutex per device serialises the ENQUEUE of factorisations (shared streams, event pool); the
// enqueued work of successive calls is ordered by the streams themselves.
namespace {
struct Aux {
  std::recursive_mutex mu;
  bool ready = false;
  int init_rc = 0;  // sticky: a failed stream set-up is reported by every later call instead of being retried
  hipStream_t P = nullptr, B = nullptr, Bs = nullptr, X = nullptr, pad = nullptr;
  hipEvent_t* ev = nullptr;
  int nev = 0;
  int ncu = 0, bulk_cus = 0;
  // stream-layout self-check (aux_get): microseconds per cross-stream hand-off P<->X, P<->Bs, X<->Bs as first measured and
  // after a possible re-creation of the streams; recreated = 1 if the first layout failed the check
  double check_us[3] = {0, 0, 0}, check_first_us[3] = {0, 0, 0};
  int recreated = 0;
  hipStream_t shift = nullptr;   // (kept alive: the extra stream that moved the re-created set onto other hardware queues)
  // packet-free hand-offs of the latency chain (potrf_core, "chain flags"): one word per panel for "panel solved" (F) and for
  // "rest-update done" (R), written with the epoch of the factorisation that owns them (monotonic per device)
  int* flags = nullptr;
  int* cnt = nullptr;    // two counters per panel for the fused panel kernel (producers done / workgroups through the solve), zeroed per call
  int epoch = 0;
  int concurrent = -1;   // 1: kernels of two streams were seen running at the same time (init-time probe); 0: serialised by a tool
};
constexpr int kMaxFlagPanels = 512;
Aux g_aux[16];

int masked_stream(hipStream_t* out, int ncu, int first, int last) {  // CUs [first, last)
  if (first <= 0 && last >= ncu) return (int)hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  uint32_t mask[32] = {0};
  for (int i = first; i < last; ++i) mask[i >> 5] |= 1u << (i & 31);
  return (int)hipExtStreamCreateWithCUMask(out, (uint32_t)((ncu + 31) / 32), mask);
}

int aux_create(Aux& a, int dev) {
  int lo = 0, hi = 0;
  GPK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipDeviceProp_t prop;
  GPK_HIP(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  a.ncu = ncu;
  // Stream -> hardware queue -> microengine pipe.  Two facts measured on MI355X (rocprofv3 kernel timelines of the
  // SVGP step, profiles/r02_*): (1) HIP keeps a pool of GPU_MAX_HW_QUEUES hardware queues per priority level: a new
  // stream opens a new queue while the pool is not full, afterwards it shares the queue with the fewest streams
  // (ties: the most recently opened queue); CU-masked and non-default-priority streams get queues of their own.
  // (2) Hardware queues are spread round-robin over FOUR pipes in creation order, and two queues of one pipe that
  // are active at the same time slow each other down badly: every kernel start / cross-queue event hand-off on them
  // then takes ~50 us instead of ~5 (queues 1 and 5, or 2 and 6: the step went from 2.2 to 3.3 - 4.4 ms).
  // Hence this creation order -- default stream = queue 1 (pipe 0) exists already:
  //   P -> queue 2 (pipe 1);  X -> queue 3 (pipe 2);  one unused stream, then Bs: with the usual pool of 2 the unused
  //   one shares X's queue and Bs lands on the default stream's (idle) queue 1, with a pool of 4 they open queues 4
  //   and 5 (pipes 3 and 0);  then the masked B -> pipe 3 (or 1).
  // The chain (P), its rest-updates (Bs) and the bulk stream (X or B) are then always on three different pipes.
  GPK_HIP(hipStreamCreateWithPriority(&a.P, hipStreamNonBlocking, hi));
  GPK_HIP(hipStreamCreateWithFlags(&a.X, hipStreamNonBlocking));
  GPK_HIP(hipStreamCreateWithFlags(&a.pad, hipStreamNonBlocking));
  GPK_HIP(hipStreamCreateWithFlags(&a.Bs, hipStreamNonBlocking));
  int reserved = GPK_TUNE(RESERVED_CUS, 32);   // (8 until round 6: see the tile queue of the trailing updates, potrf_core)
  if (ncu > 1024 || reserved < 0 || reserved >= ncu) reserved = 0;
  int rc = masked_stream(&a.B, ncu, reserved, ncu);
  if (rc) return rc;
  a.bulk_cus = ncu - reserved;
  return 0;
}

// Cost of one cross-stream hand-off (kernel on a -> event -> kernel on b -> event -> ...), microseconds: ~5 when the two
// hardware queues sit on different microengine pipes, ~50 when they share one.
int handoff_us(hipStream_t a, hipStream_t b, double* us) {
  const int n = 24;
  hipEvent_t e0, e1, ea, eb;
  GPK_HIP(hipEventCreate(&e0));
  GPK_HIP(hipEventCreate(&e1));
  GPK_HIP(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
  GPK_HIP(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
  int rc = 0;
  for (int rep = 0; rep < 2 && !rc; ++rep) {  // (first repetition warms the queues up)
    GPK_HIP(hipEventRecord(e0, a));
    for (int i = 0; i < n && !rc; ++i) {
      rc = gpk_launch_noop(a);
      if (!rc) rc = (int)hipEventRecord(ea, a);
      if (!rc) rc = (int)hipStreamWaitEvent(b, ea, 0);
      if (!rc) rc = gpk_launch_noop(b);
      if (!rc) rc = (int)hipEventRecord(eb, b);
      if (!rc) rc = (int)hipStreamWaitEvent(a, eb, 0);
    }
    if (!rc) rc = (int)hipEventRecord(e1, a);
    if (!rc) rc = (int)hipEventSynchronize(e1);
  }
  float ms = 0.f;
  if (!rc) rc = (int)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
  *us = (double)ms * 1e3 / (2.0 * n);
  return rc;
}

// Per-kernel cost (microseconds) of n empty kernels on EACH of the given streams while all of them are being fed at the
// same time -- two active hardware queues on one microengine pipe show up here ("pipes").
int concurrent_us(hipStream_t* st, int ns, double* us) {
  const int n = 48;
  hipEvent_t e0[4], e1[4];
  for (int i = 0; i < ns; ++i) {
    GPK_HIP(hipEventCreate(&e0[i]));
    GPK_HIP(hipEventCreate(&e1[i]));
  }
  int rc = 0;
  for (int rep = 0; rep < 2 && !rc; ++rep) {
    for (int i = 0; i < ns && !rc; ++i) rc = (int)hipEventRecord(e0[i], st[i]);
    for (int k = 0; k < n && !rc; ++k)
      for (int i = 0; i < ns && !rc; ++i) rc = gpk_launch_noop(st[i]);
    for (int i = 0; i < ns && !rc; ++i) rc = (int)hipEventRecord(e1[i], st[i]);
    for (int i = 0; i < ns && !rc; ++i) rc = (int)hipEventSynchronize(e1[i]);
  }
  for (int i = 0; i < ns; ++i) {
    float ms = 0.f;
    if (!rc) rc = (int)hipEventElapsedTime(&ms, e0[i], e1[i]);
    us[i] = (double)ms * 1e3 / n;
    (void)hipEventDestroy(e0[i]);
    (void)hipEventDestroy(e1[i]);
  }
  return rc;
}

// (caller holds a.mu)
int aux_get(int dev, int need, Aux** out) {
  Aux& a = g_aux[dev];
  if (!a.ready) {
    if (a.init_rc) return a.init_rc;
    const int rc = aux_create(a, dev);
    if (rc) {  // no half-built stream set: give back what was created, remember the error
      for (hipStream_t* s : {&a.P, &a.X, &a.pad, &a.Bs, &a.B}) {
        if (*s) (void)hipStreamDestroy(*s);
        *s = nullptr;
      }
      a.init_rc = rc;
      return rc;
    }
    // Init-time self-check of the stream -> hardware-queue -> pipe layout ("pipes"): two of the three concurrently
    // active queues on one microengine pipe cost ~50 us per cross-stream hand-off instead of ~5-15, and a whole process then
    // runs 10-17 % slow at every problem size (seen on 2 of ~12 boxes in round 3).  Measured once here (a few hundred empty
    // kernels, ~1 ms, the only place the library synchronises); if any pair is slow the streams are created again behind
    // one more placeholder stream -- which shifts every stream of the set to the next hardware queue -- and measured again.
    {
      auto measure = [&](double* us) {
        (void)handoff_us(a.P, a.X, &us[0]); (void)handoff_us(a.P, a.Bs, &us[1]); (void)handoff_us(a.X, a.Bs, &us[2]);
      };
      measure(a.check_us);
      for (int i = 0; i < 3; ++i) a.check_first_us[i] = a.check_us[i];
      const double limit = (double)GPK_TUNE(HANDOFF_LIMIT_US, 30);
      if (a.check_us[0] > limit || a.check_us[1] > limit || a.check_us[2] > limit) {
        for (hipStream_t* st : {&a.P, &a.X, &a.pad, &a.Bs, &a.B}) {
          if (*st) (void)hipStreamDestroy(*st);
          *st = nullptr;
        }
        (void)hipStreamCreateWithFlags(&a.shift, hipStreamNonBlocking);
        const int rc2 = aux_create(a, dev);
        if (rc2) { a.init_rc = rc2; return rc2; }
        a.recreated = 1;
        measure(a.check_us);
      }
    }
    if (kGpkExp && GPK_TUNE(STREAM_SELFTEST, 0)) {
      double pm = 0;
      (void)handoff_us(a.P, a.B, &pm);
      hipStream_t trio[3] = {a.P, a.X, a.Bs};
      double cu[3] = {0, 0, 0};
      (void)concurrent_us(trio, 3, cu);
      fprintf(stderr, "[gpk] stream hand-off us: P<->X %.1f  P<->Bs %.1f  X<->Bs %.1f  P<->B(masked) %.1f (first layout %.1f %.1f %.1f, recreated %d) | concurrent noop us/kernel: P %.1f X %.1f Bs %.1f\n",
              a.check_us[0], a.check_us[1], a.check_us[2], pm, a.check_first_us[0], a.check_first_us[1], a.check_first_us[2], a.recreated, cu[0], cu[1], cu[2]);
    }
    a.ready = true;
  }
  if (!a.flags) {
    GPK_HIP(hipMalloc((void**)&a.flags, sizeof(int) * (4 * kMaxFlagPanels + 8)));   // F, R, the fused panels' counters, the x_tail words
    GPK_HIP(hipMemset(a.flags, 0, sizeof(int) * (4 * kMaxFlagPanels + 8)));
    a.cnt = a.flags + 2 * kMaxFlagPanels;
    // in-kernel hand-offs need kernels of two streams to RUN concurrently: under rocprofv3 --pmc (or any tool that serialises
    // kernels) they would deadlock, so the chain then keeps its events (gpk_probe_concurrent_kernels: <= 2 ms, once per device)
    int conc = 0;
    const int rcp = gpk_probe_concurrent_kernels(a.X, a.P, a.flags, &conc);
    a.concurrent = (rcp == 0 && conc) ? 1 : 0;
  }
  if (a.nev < need) {
    hipEvent_t* n = (hipEvent_t*)realloc(a.ev, sizeof(hipEvent_t) * need);
    if (!n) return GPK_E_ARG;
    a.ev = n;
    // The events only order streams of ONE device against each other (never inspected from the host), so they carry no
    // system-scope fence: the producing kernels' own release at the end of their dispatch makes the data visible to the
    // device.  Round 3, same box: chain of n = 2048 alone 0.98 -> 0.925 ms, SVGP step 2.085 -> 2.063 ms, GPR N = 16384
    // 32.75 -> 32.45 ms, the 1024-row rank shard 1.447 -> 1.388 ms, all bit-identical (profiles/r03_ab_svgp_schedules.log).
    // (Round 1 had measured this flag slower on a different schedule: 283 vs 308 steps/s.)
    for (int i = a.nev; i < need; ++i) {
      GPK_HIP(hipEventCreateWithFlags(&a.ev[i], hipEventDisableTiming | (GPK_TUNE(EV_NOFENCE, 1) ? hipEventDisableSystemFence : 0)));
      a.nev = i + 1;
    }
  }
  *out = &a;
  return 0;
}

int current_device(int* dev) {
  GPK_HIP(hipGetDevice(dev));
  if (*dev < 0 || *dev >= 16) return GPK_E_UNSUPPORTED;
  return 0;
}

// factor the outer panel [c0,c1) of the square part (rows up to `rows`) on stream s
int factor_panel(hipStream_t s, double* A, int rows, int c0, int c1, long lda, int batch, long strideA,
                 double* invd, long strideInv, int* info, int chain_wgs = 0, int chain_kparts = 0) {
  int rc;
  for (int j0 = c0; j0 < c1; j0 += NB) {
    const int j1 = (j0 + NB < c1) ? j0 + NB : c1;
    const int nb = j1 - j0;
    double* invb = invd + (long)(j0 / NB) * NB * NB;
    rc = gpk_launch_leaf(s, A + (long)j0 * lda + j0, lda, strideA, nb, invb, strideInv, info, j0, batch, 0);
    if (rc) return rc;
    const int below = rows - j1;
    if (below <= 0) continue;
    double* panel = A + (long)j1 * lda + j0;
    // in-place panel solve X = panel * inv(L11)^T: one column tile, so each workgroup only
    // overwrites rows that it alone has read
    GemmArgs g = gemm_base(below, nb, nb, 1.0, panel, lda, invb, NB, 0.0, panel, lda, batch, strideA,
                           strideInv, strideA);
    g.b_tri = 2;
    g.max_wgs = chain_wgs;
    g.small_kparts = chain_kparts;
    rc = gpk_launch_gemm(s, g);
    if (rc) return rc;
    const int ncols = c1 - j1;
    if (ncols > 0) {
      GemmArgs u = gemm_base(below, ncols, nb, -1.0, panel, lda, panel, lda, 1.0,
                             A + (long)j1 * lda + j1, lda, batch, strideA, strideA, strideA);
      u.c_lower = 1;
      u.small_kparts = chain_kparts;
      rc = gpk_launch_gemm(s, u);
      if (rc) return rc;
    }
  }
  return 0;
}

// Rows E [rows, n] against the finished columns [c0,c1) of the factor L (ldl), right-looking:
//   S[:,c0:c1] = E[:,c0:c1] L[c0:c1,c0:c1]^-T             (NB-blocked, diagonal-block inverses; after block j is solved
//                                                          ALL remaining columns of the group get one K = 128 update --
//                                                          the left-looking form was latency-bound at 44 / 58 / 74 us)
//   E[:,c1:n] -= S[:,c0:c1] L[c1:n,c0:c1]^T                (one large GEMM, K = c1 - c0)
// The solved columns S are written to Eo (ldeo) -- the same matrix as E for the in-place form, a separate one when the
// caller wants A^T apart from the consumed input rows.  Used for the extra rows of the factorisation and, group after
// group, by gpk_trsm(trans = 0).
// part_j0 >= 0 (progressive form, first group of an SVGP-size factorisation): only leaf block part_j0 of the group is solved by
// this call -- it has just been factored -- and the group's later blocks get its K = 128 update; the large GEMM follows the LAST block.
bool group_solve_fused_ok(int nbk, int c0, int c1, int rows, const double* L, long ldl, const double* invd, int batch, long strideL,
                          long strideInv) {
  const bool batch_ok = batch <= 1 || (GPK_TUNE(GROUP_FUSED_BATCH, 1) && !(strideL & 1) && !(strideInv & 1));
  return batch_ok && nbk >= 2 && nbk <= 4 && nbk * NB == c1 - c0 && (c0 % NB) == 0 && rows >= GPK_TUNE(GROUP_FUSED_MIN_ROWS, 1024) &&
         !(ldl & 1) && !(reinterpret_cast<uintptr_t>(L + (long)c0 * ldl + c0) & 15) &&
         !(reinterpret_cast<uintptr_t>(invd + (long)(c0 / NB) * NB * NB) & 15) && GPK_TUNE(GROUP_FUSED, 1);
}

int solve_group_fwd(hipStream_t s, const Bulk& bulk, double* E, long lde, double* Eo, long ldeo, int rows, const double* L,
                    long ldl, const double* invd, long strideInv, int n, int c0, int c1, int batch, long strideE,
                    long strideEo, long strideL, int part_j0 = -1, int part_cap = 0) {
  int rc;
  const int nbk = (c1 - c0) / NB;
  if (part_j0 >= 0) {
    rc = gpk_launch_group_solve(s, E + c0, lde, Eo + c0, ldeo, rows, L + (long)c0 * ldl + c0, ldl, invd + (long)(c0 / NB) * NB * NB,
                                nbk, batch, strideE, strideEo, strideL, strideInv, part_cap, part_j0, part_j0 + 1);
    if (rc) return rc;
    if (part_j0 + 1 < nbk) return 0;
  } else
  // (a batch of problems: blockIdx.y walks them; C5 with separate kernels 2.14 -> 2.07 ms against the tiled per-block launches)
  if (group_solve_fused_ok(nbk, c0, c1, rows, L, ldl, invd, batch, strideL, strideInv)) {
    // (the fused kernel stages its operand tiles by 16-byte LDS-DMA: an 8-byte-aligned factor takes the per-block loop below)
    // the whole in-group phase (nbk solves + nbk - 1 updates of the latency kernel) as ONE launch with the same arithmetic
    rc = gpk_launch_group_solve(s, E + c0, lde, Eo + c0, ldeo, rows, L + (long)c0 * ldl + c0, ldl, invd + (long)(c0 / NB) * NB * NB,
                                nbk, batch, strideE, strideEo, strideL, strideInv, bulk.group_cap);
    if (rc) return rc;
  } else {
    for (int j0 = c0; j0 < c1; j0 += NB) {
      const int j1 = (j0 + NB < c1) ? j0 + NB : c1;
      const int nb = j1 - j0;
      GemmArgs g = gemm_base(rows, nb, nb, 1.0, E + j0, lde, invd + (long)(j0 / NB) * NB * NB, NB, 0.0,
                             Eo + j0, ldeo, batch, strideE, strideInv, strideEo);
      g.b_tri = 2;
      bulk.apply(g);
      rc = gpk_launch_gemm(s, g);
      if (rc) return rc;
      if (j1 < c1) {
        GemmArgs u = gemm_base(rows, c1 - j1, nb, -1.0, Eo + j0, ldeo, L + (long)j1 * ldl + j0, ldl, 1.0,
                               E + j1, lde, batch, strideEo, strideL, strideE);
        bulk.apply(u);
        // K = 128 updates inside a group: the one-shot latency kernel with its workgroups walking the row blocks (B tile
        // staged once) instead of the tiled kernel, which runs K = 128 at 16-24 TFLOP/s (33-45 us per launch at 8192 rows).
        // Same-box A/B: SVGP step 2.251 -> 2.222 ms, GPR predict 56.5 -> 56.0 ms, cached posterior 20.9 -> 20.7 ms
        // (512 workgroups; 256: 2.238, 768: 2.246).
        if (GPK_TUNE(XSMALL, 1) && batch <= 1) {
          u.small_loop = 1;
          u.max_wgs = GPK_TUNE(XSMALL_WGS, 512);
        }
        rc = gpk_launch_gemm(s, u);
        if (rc) return rc;
      }
    }
  }
  if (c1 < n) {
    GemmArgs u = gemm_base(rows, n - c1, c1 - c0, -1.0, Eo + c0, ldeo, L + (long)c1 * ldl + c0, ldl, 1.0,
                           E + c1, lde, batch, strideEo, strideL, strideE);
    // (round 6, late: this update on the CU-masked stream B with TWO persistent workgroups per compute unit of its mask -- K loop at 88 %
    //  instead of 79 %, the 32 CUs outside the mask free for the chain, hand-over by events -- makes every SVGP workload 10 - 20 % SLOWER:
    //  Cm 1.94 - 1.96 against 1.75 - 1.79 ms, profiles/r06_ab_xbulk_masked.log.  A third active hardware queue, as in rounds 2 - 3.)
    bulk.apply(u);
    rc = gpk_launch_gemm(s, u);
    if (rc) return rc;
  }
  return 0;
}

// The mirror image for  B <- B L^-1  with LT = L^T (upper, row-major) and the transposed block inverses: columns
// [c0,c1) are solved from the last block to the first, then ONE K = c1 - c0 update of all columns to their left.
int solve_group_bwd(hipStream_t s, double* Bm, long ldb, int rows, const double* LT, long ldl, const double* invdT,
                    long strideInv, int c0, int c1, int batch, long strideB, long strideL) {
  int rc;
  for (int j1 = c1; j1 > c0;) {
    const int j0 = (j1 - c0 > NB) ? c0 + ((j1 - c0 - 1) / NB) * NB : c0;
    const int nb = j1 - j0;
    GemmArgs g = gemm_base(rows, nb, nb, 1.0, Bm + j0, ldb, invdT + (long)(j0 / NB) * NB * NB, NB, 0.0, Bm + j0,
                           ldb, batch, strideB, strideInv, strideB);
    g.b_tri = 1;
    rc = gpk_launch_gemm(s, g);
    if (rc) return rc;
    if (j0 > c0) {  // B[:, c0:j0] -= X[:, j0:j1] (LT[c0:j0, j0:j1])^T
      GemmArgs u = gemm_base(rows, j0 - c0, nb, -1.0, Bm + j0, ldb, LT + (long)c0 * ldl + j0, ldl, 1.0, Bm + c0,
                             ldb, batch, strideB, strideL, strideB);
      rc = gpk_launch_gemm(s, u);
      if (rc) return rc;
    }
    j1 = j0;
  }
  if (c0 > 0) {  // B[:, 0:c0] -= X[:, c0:c1] (LT[0:c0, c0:c1])^T
    GemmArgs u = gemm_base(rows, c0, c1 - c0, -1.0, Bm + c0, ldb, LT + c0, ldl, 1.0, Bm, ldb, batch, strideB,
                           strideL, strideB);
    rc = gpk_launch_gemm(s, u);
    if (rc) return rc;
  }
  return 0;
}

// p_prologue: work of the CALLER that the first leaf waits for and nothing else does -- the fused drivers' Kuu build.  It is
// enqueued ON the panel stream, so the first leaf follows it back to back (0.3 us) instead of behind an event record on the caller's
// stream and a wait on the panel stream (~15 us per step, round 5).
// b_prologue (large factorisations): work of the CALLER that everything EXCEPT the first panel's columns waits for -- the GPR
// driver builds only those columns before the call and the rest of K(X, X) here, on the bulk stream, beside the first panel's
// chain, which nothing else would overlap (first_panel_columns below tells it how many columns that is).
// x_prologue: work of the CALLER that belongs on the bulk stream before the first extra-row group (the SVGP driver's Kfu
// build, transposes, KL).  It is enqueued after the first panel's chain kernels: every host call issued before the first
// leaf delays the whole step, and nothing on the bulk stream is needed for ~4 panels.
// late_work: work of the CALLER that nothing in the factorisation needs (the whitened driver's tril(q_sqrt)^T and KL term).  It is
// enqueued on the rest-update stream after the sixth panel: the first four panels are HOST-bound -- ~7 enqueue calls of 5 - 8 us
// per panel against ~55 us of kernels -- so every launch issued there delays the chain (round 5: the second leaf started 52 us
// after the first strip had finished), and the rest-update stream has a leaf's time of slack per panel.
typedef std::function<int(hipStream_t)> StreamWork;
// x_tail (round 6): work of the CALLER that reads the solved extra rows and that the caller's NEXT kernel does not need -- the SVGP
// drivers' row statistics (28 us of HBM reads at Cm) beside the projection GEMM.  If the factorisation hands its extra rows over with
// flag words (small sizes, gate kernels) it enqueues the work on the extra-row stream right behind the last solve, publishes a second
// word behind it and reports that word: the caller's consumer waits for it in-kernel (VarexpExtra).  The caller's stream is released by a
// one-wave gate on the first word (~1 us behind the solve; the event pair it replaces cost ~10 us).  Otherwise `used` stays false and
// the caller runs the work itself.  MEASURED (profiles/r06_ab_x_tail.log, two repetitions per setting on one box): no gain -- Cm 1.79 / 1.80 ms
// without / with it, C3 0.70 / 0.73: the 2048 light workgroups of the statistics take the wave slots the projection's first tiles want and
// the projection ends as much later as it started earlier.  Likewise the slot partials summed inside the variational-expectation kernel
// instead of a sum_parts launch (level).  Both are A/B knobs (GPK_XTAIL, GPK_VAREXP_SUMS_PARTS), off.
struct XTail {
  const StreamWork* work = nullptr;
  bool used = false;
  const int* done_ptr = nullptr;
  int done_val = 0;
};
struct PotrfHooks {
  const StreamWork* x_prologue = nullptr;
  const StreamWork* b_prologue = nullptr;
  const StreamWork* p_prologue = nullptr;
  const StreamWork* late_work = nullptr;
  XTail* x_tail = nullptr;
};

int potrf_core(hipStream_t S, double* A, int n, int extra, long lda, int batch, long strideA, double* invd, int zero_upper,
               int* info, const PotrfHooks& hooks = PotrfHooks(), int tri = 0, bool tri_prefilled = false) {
  const StreamWork* x_prologue = hooks.x_prologue;
  const StreamWork* b_prologue = hooks.b_prologue;
  const StreamWork* p_prologue = hooks.p_prologue;
  const StreamWork* late_work = hooks.late_work;
  if (!A || !invd || n < 0 || extra < 0 || lda < n) return GPK_E_ARG;
  if (tri && (tri != n || extra < n || batch > 1)) return GPK_E_ARG;
  if (batch <= 0) batch = 1;
  // (the status word is reset by the leaf of column 0 -- leaf2_device.h -- not by a memset packet ahead of the fork below)
  if (n == 0) {
    if (info) GPK_HIP(hipMemsetAsync(info, 0, sizeof(int) * batch, S));
    return 0;
  }
  // tri = n: the LAST n extra rows are the identity (written here) and come back as L^-T.  Row j of that block stays
  // zero left of column j, so column group [c0, c1) only has to process its first c1 rows: n^3 / 3 flop instead of n^3.
  // tri_prefilled: the caller (or its x_prologue) puts an UPPER-TRIANGULAR block there itself -- tril(q_sqrt)^T of the
  // un-whitened ELBO: the same rows-stay-zero argument holds for any block that is zero left of its diagonal.
  if (tri && !tri_prefilled) {
    const int rci = gpk_launch_set_identity(S, A + (long)(n + extra - tri) * lda, n, lda);
    if (rci) return rci;
  }
  const long strideInv = (long)gpk_cdiv(n, NB) * NB * NB;
  // outer panel width for n >= 4096 (A/B at N = 16384, profiles/r03_ab_gpr_nbo.log); one leaf block for the SVGP sizes,
  // where the whole factorisation is a latency chain
  const int nbo_large = (GPK_TUNE(NBO, 640) / NB) * NB;
  const int nbo = (n >= 4096) ? (nbo_large >= NB ? nbo_large : NBO) : NB;
  // Panel boundaries.  The END of a large factorisation is a latency chain again (trailing matrix too small to hide the
  // panel): there a wide panel costs 5 leaves + 4 in-panel updates + one K = 640 look-ahead strip of < 256 tiles, i.e. ONE
  // under-filled tile time of ~170 us -- 450 - 480 us per 640 columns (in-kernel time stamps, tools/leaf_phase_probe.py) --
  // while single-leaf panels cost 56 - 63 us each once their K = 128 rest-updates keep up.  So the last `narrow_tail`
  // columns are factored with the SVGP-size scheme (nbo = NB).  A/B at N = 16384, same box (profiles/r03_ab_gpr_nbo.log):
  // off 32.7 ms, 2048 -> 32.65, 3072 -> 32.4, 4096 -> 31.9, 5120 -> 32.2, 6144 -> 32.5, 8192 -> 33.2.
  const int narrow_tail = (nbo > NB) ? (GPK_TUNE(NARROW_TAIL, 4096) / NB) * NB : 0;
  std::vector<int> cuts;
  for (int c = 0; c < n;) {
    cuts.push_back(c);
    c += (nbo > NB && n - c > narrow_tail) ? nbo : NB;
  }
  cuts.push_back(n);
  const int npanels = (int)cuts.size() - 1;
  // Few extra rows (GPR: the P columns of Y) simply ride along through the panel solves and trailing
  // updates of the square part; many extra rows (SVGP: the minibatch; GPR: the test rows of predict_f) are solved
  // right-looking, group by group, as bulk work overlapped with the factorisation.
  const bool ride = extra > 0 && extra <= 256;
  const int R = ride ? n + extra : n;  // rows handled together with the square part
  const bool useX = extra > 0 && !ride;
  double* E = A + (long)n * lda;       // the extra rows
  int rc;
  if (n <= NB) {  // one leaf; nothing to overlap
    if (p_prologue) {
      rc = (*p_prologue)(S);
      if (rc) return rc;
    }
    if (x_prologue) {
      rc = (*x_prologue)(S);
      if (rc) return rc;
    }
    if (late_work) {
      rc = (*late_work)(S);
      if (rc) return rc;
    }
    rc = factor_panel(S, A, R, 0, n, lda, batch, strideA, invd, strideInv, info);
    if (rc) return rc;
    if (useX) {
      rc = solve_group_fwd(S, Bulk{}, E, lda, E, lda, extra, A, lda, invd, strideInv, n, 0, n, batch, strideA, strideA, strideA);
      if (rc) return rc;
    }
    return zero_upper ? gpk_launch_zero_upper(S, A, n, lda, batch, strideA) : 0;
  }
  int dev = 0;
  rc = current_device(&dev);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lock(g_aux[dev].mu);
  Aux* aux = nullptr;
  rc = aux_get(dev, 2 * npanels + 8, &aux);   // (+ fork, three joins, the b_prologue event)
  if (rc) return rc;
  const bool large = n >= 4096;
  hipStream_t P = aux->P, B = large ? aux->B : aux->Bs;
  // ONE bulk stream beside the chain: for large factorisations the extra rows share the (hardware-masked) stream of the
  // trailing updates; for small ones they have the unmasked stream X.  Round 3 re-measured every alternative on the SVGP
  // step (profiles/r03_ab_svgp_schedules.log): a masked extra-row stream with 8 ... 128 reserved CUs, two row halves on
  // two streams, the projection streamed or split onto a side stream, one GEMM per column group against an explicit
  // group inverse -- each 5 ... 40 % slower than this scheme.
  hipStream_t X = large ? aux->B : aux->X;
  Bulk bulk;
  // cap on the persistent workgroups of the big extra-row updates, so that some CUs stay free for the panel stream's
  // one-shot kernels (A/B on the SVGP step, round 1: cap 320 -> 448 steps/s, no cap 435, cap 224 -> 431; round 3: 256 ->
  // 419, 320 -> 441, 384 -> 447)
  // (round 5, with the packet-free chain: 224 -- one workgroup on 224 compute units, 32 left to the chain's one-shot kernels --
  //  is level with 320 on the whitened step and 2 - 5 % faster on the un-whitened one, whose extra-row stream is a quarter
  //  longer; a batch of problems keeps 320: C5 separate 2.04 against 2.02 ms; 240 / 248 lose 5 %, profiles/r05_ab_caps.log)
  // (round 6 EXPERIMENT, off.)  With many extra rows the chain's one-shot kernels can stage K in two halves (gemm_nt_small, kparts = 2:
  // 74 KB of LDS) so that they fit BESIDE a capped bulk workgroup (84 KB) on the same compute unit instead of queueing through the few
  // CUs the cap leaves free (three rounds of ~10 us per launch while an update holds 224 CUs, profiles/r06_step_timeline.txt), and the
  // cap could then go up.  Measured, same box (profiles/r06_ab_halfk.log): Cm 1.76 ms without, 1.80 with it at the same cap of
  // 224, 1.85 / 1.88 at caps of 240 / 254 -- the second staging round trip costs more than the queueing, and more bulk workgroups
  // slow the extra-row stream itself.  Kept as an A/B knob (GPK_CHAIN_HALFK=1).
  const bool chain_halfk = useX && !large && batch == 1 && nbo == NB && extra >= GPK_TUNE(CHAIN_HALFK_MIN_ROWS, 6144) &&
                           GPK_TUNE(CHAIN_HALFK, 0);
  const int chain_kparts = chain_halfk ? 2 : 0;
  if (!large) bulk.cap = batch > 1 ? GPK_TUNE(EXTRA_MAX_WGS_BATCHED, 320)
                                   : (chain_halfk ? GPK_TUNE(EXTRA_MAX_WGS_HALFK, 248) : GPK_TUNE(EXTRA_MAX_WGS, 224));
  if (large && GPK_TUNE(XQUEUE_LARGE, 0)) bulk.queue_cus = aux->bulk_cus;   // (A/B, off: GPR predict's test-row updates, level at 54.4 - 55.0 ms)
  if (!large) bulk.group_cap = GPK_TUNE(GROUP_SOLVE_MAX_WGS, 0);
  if (!large) bulk.kmin = GPK_TUNE(EXTRA_CAP_KMIN, 256);
  hipEvent_t* evF = aux->ev;            // [npanels] panel p factored, rows below solved (recorded on P)
  hipEvent_t* evR = aux->ev + npanels;  // [npanels] rest of the trailing update of panel p done (on B)
  hipEvent_t evFork = aux->ev[2 * npanels], evJoinP = aux->ev[2 * npanels + 1], evJoinB = aux->ev[2 * npanels + 2],
             evJoinX = aux->ev[2 * npanels + 3];
  const bool p_on_panel = p_prologue && GPK_TUNE(KUU_ON_PANEL, 1);
  if (p_prologue && !p_on_panel) {
    rc = (*p_prologue)(S);
    if (rc) return rc;
  }
  if (x_prologue && !useX) {  // the extra rows ride through the panel solves: they must exist before the first one
    rc = (*x_prologue)(S);
    if (rc) return rc;
  }
  GPK_HIP(hipEventRecord(evFork, S));  // fork: everything already queued on S comes first
  GPK_HIP(hipStreamWaitEvent(P, evFork, 0));
  if (B != S) GPK_HIP(hipStreamWaitEvent(B, evFork, 0));
  if (useX && X != B) GPK_HIP(hipStreamWaitEvent(X, evFork, 0));
  if (p_on_panel) {
    rc = (*p_prologue)(P);
    if (rc) return rc;
  }
  // (round 6 EXPERIMENT, off: the fused panel kernel -- solve + strip in one launch, gemm.hip -- is bit-identical and level with the two
  //  launches it replaces: 1024-row shard 0.976 against 0.985 ms, Cm 1.795 / 1.79, C3 0.766 / 0.764 (profiles/r06_ab_panel_fused.log).  What
  //  it saves in launch ramp and re-staging it spends on the cross-workgroup hand-over of the B tile through memory.  A/B: GPK_PANEL_FUSED=1.)
  const bool panel_fused_on = GPK_TUNE(PANEL_FUSED, 0) && batch == 1 && aux->cnt != nullptr;
  int* pending_sig = nullptr;   // "panel solved" word of a fused panel that the NEXT kernel of the panel stream still has to announce
  if (panel_fused_on)   // the fused panel kernels' counters (everything of earlier calls that used them has completed: P waited for the fork)
    GPK_HIP(hipMemsetAsync(aux->cnt, 0, sizeof(int) * 2 * (size_t)std::min(npanels, kMaxFlagPanels), P));
  hipStream_t last_bulk = B;
  int last_rest = -1;  // panel index whose evR marks the most recent rest-update
  hipEvent_t evBpro = aux->ev[2 * npanels + 4];
  bool bpro_pending = false;
  if (b_prologue) {
    rc = (*b_prologue)(B);
    if (rc) return rc;
    if (B != S) {
      GPK_HIP(hipEventRecord(evBpro, B));
      bpro_pending = true;
      if (useX && X != B) GPK_HIP(hipStreamWaitEvent(X, evBpro, 0));
      if (aux->Bs != B) GPK_HIP(hipStreamWaitEvent(aux->Bs, evBpro, 0));
    }
  }
  const bool use_flags = GPK_TUNE(CHAIN_FLAGS, 1) && (batch == 1 || GPK_TUNE(CHAIN_FLAGS_BATCHED, 1)) && aux->flags != nullptr &&
                         aux->concurrent == 1;
  int* flagF = aux->flags;
  int* flagR = aux->flags + kMaxFlagPanels;
  const bool gate_kernels = GPK_TUNE(GATE_KERNELS, 1) != 0;
  const int epoch = ++aux->epoch;
  bool rest_flagged = false;   // the most recent rest-update was followed by a write of R[last_rest]
  std::vector<char> panel_flagged(npanels, 0);
  int xg0 = 0;         // first column of the current extra-row group
  // (512 columns for M = 2048: 256 / 384 measured slower there.  For M <= 1024 the extra-row stream would start after half of
  // the chain: 256 columns for a batch of problems -- C5 separate 2.036 -> 1.977 ms -- and 128 for a single one -- C3 0.834 ->
  // 0.803 ms, C5 shared 1.314 -> 1.30 ms, but C5 separate 1.97 -> 2.11; profiles/r04_ab_c5.log, r04_ab_xgroup_small.log)
  // (round 6, with the chain at 41 us per panel instead of 62 the extra-row stream is the longer of the two at M = 1024 and wider groups
  //  -- fewer, longer-K updates of its 8192 rows -- win: 128 / 256 / 384 / 512 columns: C3 0.757 / 0.713 / 0.688 / 0.716 ms, C5 shared
  //  1.24 / 1.20 / 1.19 / 1.21 ms, two repetitions each on one box, profiles/r06_ab_extra_row_groups.log)
  const int xgroup_small = batch > 1 ? GPK_TUNE(XGROUP_SMALL_BATCH, 256) : GPK_TUNE(XGROUP_SMALL, 384);
  // (round 6: with FEW extra rows -- a rank's shard of a strong-scaled step -- M = 2048 prefers 256-column groups too: 4096 / 2048 / 1024
  //  rows 1.353 / 1.049 / 0.938 -> 1.308 / 1.019 / 0.918 ms, while 8192 rows lose 5 %: tools/strong_scaling_emulation.py under GPK_XGROUP,
  //  profiles/r06_ab_extra_row_groups.log)
  const int xgroup_wide = (batch == 1 && n < 4096 && extra < GPK_TUNE(XGROUP_FEW_ROWS_BELOW, 6144)) ? GPK_TUNE(XGROUP_FEW_ROWS, 256) : GPK_TUNE(XGROUP, NBO);
  const int xgroup = std::max(NB, ((n <= 1024 ? xgroup_small : xgroup_wide) / NB) * NB);
  // (round 5 knobs: width of the FIRST extra-row group -- the extra-row stream idles until it is factored -- and the row count above
  //  which the shrinking groups at the end are dropped: with many rows that stream, not the chain, finishes last)
  const int xgroup_first = std::max(NB, (GPK_TUNE(XGROUP_FIRST, 0) > 0 ? (GPK_TUNE(XGROUP_FIRST, 0) / NB) * NB : xgroup));
  // (A/B, profiles/r05_ab_extra_row_stream.log: M = 2048 x 8192 rows 1.97 - 1.99 -> 1.934 ms without the shrinking groups;
  //  M = 1024, whose every panel is a group already, keeps them: 0.76 against 0.78 ms)
  // (round 6, late: with the 41-us chain period the extra-row stream finishes last at M = 1024 too -- the two single-block groups at the end ran as
  //  three launches BEHIND the last leaf: C3 0.683 - 0.690 -> 0.662 - 0.678 ms, C5 shared 1.18 -> 1.15, profiles/r06_ab_tail_zone_small.log)
  const int tail_zone_max_rows = n > 1024 ? GPK_TUNE(XTAIL_ZONE_MAX_ROWS, 6144) : GPK_TUNE(XTAIL_ZONE_MAX_ROWS_SMALL, 6144);
  // (A/B, profiles/r05_ab_extra_row_stream.log: latency kernel everywhere 1.903 1.907 | tiled from 150 workgroups 1.867 1.869 |
  //  from 250: 1.886 1.896 | always: 1.883 1.897; caps of 16 / 32 / 64 walking workgroups on the latency kernel: 2.41 / 2.06 / 1.94)
  // (all three "many extra rows" switches -- this one, the progressive first group, no shrinking groups at the end -- were measured
  //  at 8192 rows (gain) and 4096 rows (loss: 1.71 -> 1.82 ms for this one, tools/strong_scaling_emulation.py): threshold 6144)
  const bool rest_tiled = useX && !large && batch == 1 && extra >= GPK_TUNE(REST_TILED_MIN_ROWS, 3000);   // (6144 until the 64 x 64 tiles below: 4096 rows 1.33 -> 1.27 ms with them, profiles/r06_ab_rest_update_tile64.log)
  const int rest_tiled_min_wgs = n > 1024 ? GPK_TUNE(REST_TILED_MIN_WGS, 30) : GPK_TUNE(REST_TILED_MIN_WGS_SMALL, 150);
  const int rest_small_wgs = (useX && !large && batch == 1 && extra >= 6144) ? GPK_TUNE(REST_SMALL_WGS, 0) : 0;
  const int prog_end = std::min(xgroup_first, n);
  const int prog_cap = GPK_TUNE(XFIRST_PART_WGS, 128);
  const bool progressive = useX && !large && nbo == NB && batch == 1 && GPK_TUNE(XFIRST_PROGRESSIVE, 1) && GPK_TUNE(GROUP_SOLVE_V2, 1) &&
                           prog_end >= 2 * NB && extra >= GPK_TUNE(XFIRST_PROGRESSIVE_MIN_ROWS, 6144) &&
                           group_solve_fused_ok(prog_end / NB, 0, prog_end, tri ? extra - tri + prog_end : extra, A, lda, invd, batch, strideA,
                                                strideInv);
  const int late_panel = std::min(npanels - 1, GPK_TUNE(LATE_WORK_PANEL, 5));
  // the event of the most recent rest-update, recorded when first needed: its stream is in order, so a record issued later covers it
  bool evr_recorded = false;
  auto need_evr = [&]() -> int {
    if (!evr_recorded && last_rest >= 0) {
      GPK_HIP(hipEventRecord(evR[last_rest], last_bulk));
      evr_recorded = true;
    }
    return 0;
  };
  for (int p = 0; p < npanels; ++p) {
    const int c0 = cuts[p], c1 = cuts[p + 1];
    const int c2 = (p + 2 <= npanels) ? cuts[p + 2] : n;
    const bool narrow = large && (c1 - c0 <= NB) && nbo > NB;  // single-leaf panel in the chain-bound end of a large factorisation
    // ---- P: the critical path.  Panel p, then the strip = columns of panel p+1 (look-ahead) -----------
    const int chain_wgs = (!large && batch == 1) ? GPK_TUNE(CHAIN_MAX_WGS, 0) : 0;
    const bool tail_zone = !large && (nbo == NB) && (n >= 8 * NB) && (extra < tail_zone_max_rows);
    const int xgroup_now = (xg0 == 0 && !large) ? xgroup_first : xgroup;
    const bool x_waits_here = useX && (c1 == n || ((c1 - xg0) >= xgroup_now || (large && c1 - xg0 >= nbo)) ||
                                       (tail_zone && (c1 == n - 2 * NB || c1 == n - NB)));
    // Fused panel kernel (round 6): a full single-leaf panel that hands over with flags runs  leaf -> ONE kernel (panel solve +
    // strip, gemm.hip: panel_fused_kernel)  instead of  leaf -> solve -> strip.  "Panel solved" is published by the last workgroup
    // through the solve; the wait for the previous rest-update sits between the two phases.
    double* const invb_p = invd + (long)(c0 / NB) * NB * NB;
    const bool fused_panel = panel_fused_on && use_flags && p < kMaxFlagPanels && c1 < n && (c1 - c0) == NB && !bpro_pending &&
                             chain_wgs == 0 && chain_kparts == 0 && !(x_waits_here && X == aux->B) && (last_rest < 0 || rest_flagged) &&
                             gpk_panel_fused_ok(A + (long)c1 * lda + c0, lda, invb_p, R - c1, NB, c2 - c1);
    if (fused_panel) {
      // ("panel p-1 solved" of a fused predecessor rides on this leaf's entry)
      rc = gpk_launch_leaf(P, A + (long)c0 * lda + c0, lda, strideA, NB, invb_p, strideInv, info, c0, batch, 0, pending_sig, epoch);
      pending_sig = nullptr;
      if (rc) return rc;
      rc = gpk_launch_panel_fused(P, A + (long)c1 * lda + c0, lda, invb_p, A + (long)c1 * lda + c1, R - c1, c2 - c1, aux->cnt + 2 * p,
                                  nullptr, epoch, last_rest >= 0 ? flagR + last_rest : nullptr, epoch, info);
      if (rc) return rc;
      pending_sig = flagF + p;   // announced by the entry of the next kernel on the panel stream
    } else {
      if (pending_sig) {
        rc = gpk_launch_set_flag(P, pending_sig, epoch);
        pending_sig = nullptr;
        if (rc) return rc;
      }
      rc = factor_panel(P, A, R, c0, c1, lda, batch, strideA, invd, strideInv, info, chain_wgs, chain_kparts);
      if (rc) return rc;
    }
    const double* Pn = A + (long)c1 * lda + c0;  // rows c1.. of the solved panel
    GemmArgs strip{};
    if (c1 < n) {
      strip = gemm_base(R - c1, c2 - c1, c1 - c0, -1.0, Pn, lda, Pn, lda, 1.0, A + (long)c1 * lda + c1, lda, batch, strideA,
                        strideA, strideA);
      strip.c_lower = 1;
      strip.max_wgs = chain_wgs;
      strip.small_kparts = chain_kparts;
    }
    // Chain flags (round 5).  Between two kernels of the panel stream an event record costs 4.6 us and an event wait 6.3 us of
    // queue-packet processing (rocprofv3 timelines, profiles/r05_rows1024_events_timeline.txt, r05_ab_chain_flags.log); two kernels back to back start
    // 0.3 us apart.  Single-leaf panels (the SVGP sizes and the narrow tail of a large factorisation: leaf -> solve -> strip,
    // 16 - 32 times per factorisation) therefore hand over WITHOUT packets on this stream:
    //   "panel p solved"     the strip kernel stores the epoch into F[p] on entry (its predecessor, the solve, has completed
    //                        and released); the rest-update and extra-row streams wait for it with hipStreamWaitValue32;
    //   "rest-update done"   hipStreamWriteValue32(R[p]) behind the rest-update on ITS stream; the next strip's workgroups
    //                        spin on it in-kernel (normally already there: the rest-update has a leaf's time of slack).
    // (stream memory operations only on the plain streams: on the CU-masked bulk stream of large factorisations a
    // hipStreamWriteValue32 was observed to overtake the kernel queued before it -- wrong factor at n = 5000 -- so a panel whose
    // extra-row group waits on that stream keeps its event, and so does a strip whose rest-update ran there)
    const bool flagged = fused_panel || (use_flags && p < kMaxFlagPanels && c1 < n && (c1 - c0) <= NB &&
                                         gpk_gemm_takes_latency_kernel(strip) && !(x_waits_here && X == aux->B));
    panel_flagged[p] = flagged ? 1 : 0;
    if (!flagged) GPK_HIP(hipEventRecord(evF[p], P));
    if (c1 < n && !fused_panel) {
      if (bpro_pending) {   // the strip is the first kernel of the chain that leaves the first panel's columns
        GPK_HIP(hipStreamWaitEvent(P, evBpro, 0));
        bpro_pending = false;
      }
      // columns c1:c2 also received the most recent rest-update (on a bulk stream): order the two
      if (last_rest >= 0) {
        if (flagged && rest_flagged) {
          strip.wait_ptr = flagR + last_rest;
          strip.wait_val = epoch;
          strip.wait_info = info;
        } else {
          rc = need_evr();
          if (rc) return rc;
          GPK_HIP(hipStreamWaitEvent(P, evR[last_rest], 0));
        }
      }
      if (flagged) {
        strip.sig_ptr = flagF + p;
        strip.sig_val = epoch;
      }
      rc = gpk_launch_gemm(P, strip);
      if (rc) return rc;
    }
    // (what the other streams wait for: the flag word of a flagged panel, else the event)
    // (a flagged panel: our own one-wave gate kernel, 0.3 us behind its predecessor, instead of the runtime's wait packet, 5 - 7 us;
    //  never on the CU-masked stream, whose stream memory operations were seen out of order -- see above)
    auto wait_panel = [&](hipStream_t st) -> int {
      if (panel_flagged[p]) {
        if (gate_kernels) return gpk_launch_wait_flag(st, flagF + p, epoch, info);
        GPK_HIP(hipStreamWaitValue32(st, flagF + p, (uint32_t)epoch, hipStreamWaitValueGte, 0xffffffffu));
      } else GPK_HIP(hipStreamWaitEvent(st, evF[p], 0));
      return 0;
    };
    // (A/B build only, GPK_FAULT_DROP_REST_FLAG=p: the "rest-update p done" word is never written -- the next strip's bounded
    //  in-kernel wait must expire, the status word become INT_MAX and the call return instead of hanging: tests/test_gpu_handoff.py)
    const bool drop_rest_flag = kGpkExp && GPK_TUNE(FAULT_DROP_REST_FLAG, -1) == p;
    auto write_rest_flag = [&](hipStream_t st) -> int {
      if (drop_rest_flag) return 0;
      if (gate_kernels) return gpk_launch_set_flag(st, flagR + p, epoch);
      GPK_HIP(hipStreamWriteValue32(st, flagR + p, (uint32_t)epoch, 0));
      return 0;
    };
    // ---- B: rest of the outer trailing update  A[c2:, c2:] -= P[c2:] P[c2:]^T, lower tiles only --------
    // While the trailing matrix is large the factorisation is bound by these GEMMs (masked stream B); they start as soon
    // as panel p is solved.
    if (c2 < n) {
      hipStream_t Bp = narrow ? aux->Bs : B;
      const double* P2 = A + (long)c2 * lda + c0;
      GemmArgs u = gemm_base(R - c2, n - c2, c1 - c0, -1.0, P2, lda, P2, lda, 1.0,
                             A + (long)c2 * lda + c2, lda, batch, strideA, strideA, strideA);
      u.c_lower = 1;
      // (round 5) While the extra-row stream's capped updates hold 224 compute units, a rest-update on the one-shot latency kernel
      // -- up to 512 workgroups of 150 KB each -- queues through the 32 free ones for ~140 us and the chain's strips queue behind
      // it; the tiled kernel's 74-KB workgroups fit beside the capped ones.
      if (rest_tiled && (long)gpk_cdiv(u.m, 16) * gpk_cdiv(u.n, 128) >= rest_tiled_min_wgs) {
        u.no_small = 1;
        // (round 6, late) ... and there a 128 x 128 tile of the rest-update shares its compute unit with a capped MFMA-bound workgroup of the
        // extra-row stream and takes 60 - 95 us instead of 30 -- longer than the chain's period, and every strip WAITS for the previous
        // rest-update (the strips of the step timeline: 30 - 67 us, of which 8 are work).  As 64 x 64 tiles of the generic kernel (four times
        // the workgroups, 36 KB of LDS: they fit anywhere) it is short again: Cm 1.771 -> 1.750 ms, with the tiled regime from 30
        // workgroups on (M > 1024) 1.72 - 1.74; 32 x 64 and 64 x 128 tiles lose (profiles/r06_ab_rest_update_tile64.log).
        u.tile64 = GPK_TUNE(REST_TILE64, 1);
      }
      else if (rest_small_wgs > 0) { u.small_loop = 1; u.max_wgs = rest_small_wgs; }
      if (Bp == aux->B && large) {
        u.stagger_first = aux->bulk_cus;
        // persistent workgroups (two per CU of the masked stream) that walk the tile list: no workgroup launch per tile
        if (GPK_TUNE(TRAIL_PERSIST, 0)) u.max_wgs = GPK_TUNE(TRAIL_PERSIST, 0) * aux->bulk_cus;
        // (round 6) persistent workgroups -- two per compute unit of the bulk stream -- that take their tiles from a device counter
        // (gemm.hip, "Tile QUEUE"): no workgroup launch per tile and no drift between static tile lists; the kernel alone gains 7 %
        // (as dispatched 0.591 -> 0.632 of the chip's peak from 240 CUs).  They never leave their CUs, though, so the look-ahead panel no
        // longer finds gaps there and needs more CUs of its own: with 8 reserved the whole factorisation LOSES 7 % (33.0 against 30.8 ms),
        // with 32 (four per XCD) it gains 1.7 % (29.94 / 30.07 against 30.58 / 30.43 ms; 24: 32.0, 40: 31.6;
        // profiles/r06_ab_gpr_tile_queue.log).
        u.tile_queue = GPK_TUNE(TRAIL_QUEUE, 1);
      }
      // Split rest-update (round 6).  With the 25-us leaf the chain of a single-leaf panel is leaf 25 + solve 7 + strip 8 = 40 us,
      // and the rest-update stream had become the longer one: wait packet 6 + one 30-us tiled launch + write packet 7 + the
      // in-kernel wait of the next strip = 45 us per panel (profiles/r06_rows1024_new_leaf_timeline.txt).  The next strip only
      // needs the NEXT block column of the rest-update, so that column goes first, on the one-shot latency kernel (~8 us, beside
      // strip p) behind a gate on "panel p solved"; the remainder follows on the same stream and announces the
      // column on ITS entry (GemmArgs::sig_ptr) -- no packet in between, and the remainder has a whole panel period of slack.
      // ("waiting for panel p solved": the one-wave gate kernel of wait_panel.)
      const int c3 = (p + 3 <= npanels) ? cuts[p + 3] : n;
      GemmArgs ua = gemm_base(R - c2, c3 - c2, c1 - c0, -1.0, P2, lda, P2, lda, 1.0, A + (long)c2 * lda + c2, lda, batch, strideA,
                              strideA, strideA);
      ua.c_lower = 1;
      ua.small_kparts = chain_kparts;
      // In the tiled regime (many extra rows, above) the whole rest-update was ONE tiled launch that shares its compute units with the
      // extra-row stream's capped workgroups and takes 60 - 95 us there instead of 30 (profiles/r06_step_timeline.txt) -- longer than the
      // chain's own 41 us, and the next strip waited for all of it.  The same split there -- the next block column first, as 64 x 64 tiles
      // of the generic kernel (36 KB of LDS: they fit beside anything), the remainder behind it -- was measured SLOWER: Cm 1.85 against 1.79 ms
      // (latency kernel 1.84, 128 x 128 tiles 1.95; profiles/r06_ab_rest_split_tiled.log): the extra-row stream is co-critical and every extra
      // launch beside it costs more than the strip gains.  A/B knob, off.
      const int split_tiled = u.no_small ? GPK_TUNE(REST_SPLIT_TILED, 0) : 0;   // 1: 64 x 64 tiles, 2: latency kernel, 3: 128 x 128 tiles
      if (split_tiled == 1) ua.tile64 = 1;
      else if (split_tiled == 3) ua.no_small = 1;
      const bool split = GPK_TUNE(REST_SPLIT, 1) && use_flags && p < kMaxFlagPanels && Bp != aux->B && panel_flagged[p] &&
                         (!u.no_small || split_tiled) && !u.small_loop && (c2 - c1) <= NB &&
                         (split_tiled == 1 || split_tiled == 3 || gpk_gemm_takes_latency_kernel(ua)) &&
                         !(last_rest >= 0 && last_bulk != Bp);
      if (split) {
        // (the gate, not an in-kernel wait: up to 120 workgroups of 150 KB spinning from the moment they are enqueued -- a leaf and
        //  a solve before their flag -- would hold the compute units the chain and the extra-row stream need)
        rc = wait_panel(Bp);
        if (rc) return rc;
        rc = gpk_launch_gemm(Bp, ua);
        if (rc) return rc;
        if (c3 < n) {
          const double* P3 = A + (long)c3 * lda + c0;
          GemmArgs ub = gemm_base(R - c3, n - c3, c1 - c0, -1.0, P3, lda, P3, lda, 1.0, A + (long)c3 * lda + c3, lda, batch,
                                  strideA, strideA, strideA);
          ub.c_lower = 1;
          ub.no_small = u.no_small;
          ub.tile64 = u.tile64;
          if (!drop_rest_flag) {
            ub.sig_ptr = flagR + p;
            ub.sig_val = epoch;
          }
          rc = gpk_launch_gemm(Bp, ub);
          if (rc) return rc;
        } else {
          rc = write_rest_flag(Bp);
          if (rc) return rc;
        }
        rest_flagged = true;
      } else {
        if (narrow) {
          // the unmasked stream of the SVGP-size scheme.  (A stream masked to half the CUs would keep CUs free for the leaf,
          // but its hand-offs to P took ~55 us instead of ~5: 190 us per panel instead of 56, GPR N = 16384 36.6 vs 32.0 ms.)
          rc = wait_panel(Bp);
          if (rc) return rc;
          if (last_rest >= 0 && last_bulk != Bp) {
            rc = need_evr();
            if (rc) return rc;
            GPK_HIP(hipStreamWaitEvent(Bp, evR[last_rest], 0));
          }
        } else {
          rc = wait_panel(B);
          if (rc) return rc;
        }
        rc = gpk_launch_gemm(Bp, u);
        if (rc) return rc;
        rest_flagged = use_flags && p < kMaxFlagPanels && Bp != aux->B;
        if (rest_flagged) {
          rc = write_rest_flag(Bp);
          if (rc) return rc;
        } else GPK_HIP(hipEventRecord(evR[p], Bp));
      }
      evr_recorded = !rest_flagged;   // (a flagged rest-update gets its event only if somebody asks for it: need_evr)
      last_bulk = Bp;
      last_rest = p;
    }
    if (p == 0 && x_prologue && useX) {
      rc = (*x_prologue)(X);
      if (rc) return rc;
    }
    if (late_work && p == late_panel) {
      // (on the stream of the most recent rest-update, whose last event the join below waits for)
      rc = (*late_work)(last_bulk);
      if (rc) return rc;
    }
    // ---- X: the extra rows against the finished columns, in groups of up to 512 columns (so that the big
    // right-looking update is a K = 512 GEMM).  For the small sizes the groups shrink towards the end (.., n-256,
    // n-128, n): whatever is left of the extra-row work when the LAST leaf finishes is exposed latency.
    // Progressive first group (round 5).  The extra-row stream has nothing to do until the first group (four panels, ~245 us) is
    // factored, and then spends ~100 us on that group's in-group solve before its first large update can start.  Instead, as soon
    // as panel j of the first group is solved, ONE leaf block of the in-group solve runs (S_j = E_j X_j^T and the K = 128 update of
    // the group's later blocks: 4 + 3 + 2 + 1 block products), on a capped number of workgroups so that the chain -- alone on the
    // critical path there -- keeps its compute units.  When the fourth panel is done only one block product is left.
    if (progressive && xg0 == 0 && c1 <= prog_end) {
      rc = wait_panel(X);
      if (rc) return rc;
      const int xrows = tri ? extra - tri + prog_end : extra;
      rc = solve_group_fwd(X, bulk, E, lda, E, lda, xrows, A, lda, invd, strideInv, n, 0, prog_end, batch, strideA, strideA, strideA,
                           c0 / NB, prog_cap);
      if (rc) return rc;
      if (c1 == prog_end) xg0 = c1;
      continue;
    }
    const bool tail_group = tail_zone && (c1 == n - 2 * NB || c1 == n - NB);
    const bool full_group = ((c1 - xg0) >= xgroup_now || (large && c1 - xg0 >= nbo)) && !(tail_zone && c1 > n - 2 * NB && c1 < n);
    if (useX && (c1 == n || full_group || tail_group)) {
      const int g0 = xg0;
      xg0 = c1;
      rc = wait_panel(X);
      if (rc) return rc;
      // (columns [g0, c1) may span several 512-groups when the outer panel is wider than a group)
      for (int h0 = g0; h0 < c1; h0 += NBO) {
        const int h1 = std::min(h0 + NBO, c1);
        const int xrows = tri ? extra - tri + h1 : extra;  // (identity rows below column h1 are still exactly zero here)
        // (round 6, late) The in-group solve of a later group is 256 workgroups of 132 KB for ~63 us: every compute unit is taken and
        // the chain -- whose kernels all need a whole CU's LDS -- stands still for as long (leaves of 85 / 66 us in the step timeline
        // exactly beside the solves of groups 1 and 2).  Block by block (the progressive form, all of it issued here: the panels are done)
        // the chain gets a compute unit between two launches.  MEASURED: the four block launches sum to ~170 us against 63 for the fused one and
        // the extra-row stream has no such slack: Cm 1.78 -> 1.84 ms (profiles/r06_ab_group_by_blocks.log).  A/B knob, off.
        const int nbk = (h1 - h0) / NB;
        const bool by_blocks = GPK_TUNE(XGROUP_BY_BLOCKS, 0) && progressive && h0 > 0 && c1 < n && nbk >= 2 && (h1 - h0) == nbk * NB &&
                               group_solve_fused_ok(nbk, h0, h1, xrows, A, lda, invd, batch, strideA, strideInv);
        if (by_blocks) {
          for (int j = 0; j < nbk; ++j) {
            rc = solve_group_fwd(X, bulk, E, lda, E, lda, xrows, A, lda, invd, strideInv, n, h0, h1, batch, strideA, strideA, strideA, j,
                                 GPK_TUNE(XGROUP_BLOCK_WGS, 0));
            if (rc) return rc;
          }
          continue;
        }
        rc = solve_group_fwd(X, bulk, E, lda, E, lda, xrows, A, lda, invd, strideInv, n, h0, h1, batch, strideA, strideA,
                             strideA);
        if (rc) return rc;
      }
    }
  }
  // join: P has waited for every rest-update it depends on; B's last event covers the rest
  GPK_HIP(hipEventRecord(evJoinP, P));
  GPK_HIP(hipStreamWaitEvent(S, evJoinP, 0));
  if (last_bulk != S) {
    GPK_HIP(hipEventRecord(evJoinB, last_bulk));  // rest-updates are chained through evR, the last one covers all
    GPK_HIP(hipStreamWaitEvent(S, evJoinB, 0));
  }
  if (useX && X != last_bulk) {
    XTail* xt = hooks.x_tail;
    if (xt && xt->work && use_flags && gate_kernels && !large && X != S && GPK_TUNE(XTAIL, 0)) {
      int* flagE = aux->flags + 4 * kMaxFlagPanels;   // [2]: extra rows solved / tail work done
      rc = gpk_launch_set_flag(X, flagE, epoch);
      if (rc) return rc;
      rc = gpk_launch_wait_flag(S, flagE, epoch, info);
      if (rc) return rc;
      rc = (*xt->work)(X);
      if (rc) return rc;
      rc = gpk_launch_set_flag(X, flagE + 1, epoch);
      if (rc) return rc;
      xt->used = true;
      xt->done_ptr = flagE + 1;
      xt->done_val = epoch;
    } else {
      GPK_HIP(hipEventRecord(evJoinX, X));
      GPK_HIP(hipStreamWaitEvent(S, evJoinX, 0));
    }
  }
  if (zero_upper) return gpk_launch_zero_upper(S, A, n, lda, batch, strideA);
  return 0;
}
}  // namespace

namespace {
// number of leading columns the FIRST outer panel of a factorisation of size n covers (same rule as potrf_core's cuts)
int first_panel_columns(int n) {
  const int nbo_large = (GPK_TUNE(NBO, 640) / NB) * NB;
  const int nbo = (n >= 4096) ? (nbo_large >= NB ? nbo_large : NBO) : NB;
  const int narrow_tail = (nbo > NB) ? (GPK_TUNE(NARROW_TAIL, 4096) / NB) * NB : 0;
  const int w = (nbo > NB && n > narrow_tail) ? nbo : NB;
  return w < n ? w : n;
}
}  // namespace

extern "C" int gpk_stream_selfcheck(double* us_now, double* us_first, int* recreated) {
  int dev = 0;
  const int rc = current_device(&dev);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lock(g_aux[dev].mu);
  const Aux& a = g_aux[dev];
  if (!a.ready) return GPK_E_UNSUPPORTED;   // no factorisation with n > 128 has been issued on this device yet
  for (int i = 0; i < 3; ++i) {
    if (us_now) us_now[i] = a.check_us[i];
    if (us_first) us_first[i] = a.check_first_us[i];
  }
  if (recreated) *recreated = a.recreated;
  return 0;
}

extern "C" int gpk_chain_handoff_mode(void) {
  int dev = 0;
  if (current_device(&dev)) return -1;
  std::lock_guard<std::recursive_mutex> lock(g_aux[dev].mu);
  const Aux& a = g_aux[dev];
  if (!a.ready || a.concurrent < 0) return -1;
  if (!(GPK_TUNE(CHAIN_FLAGS, 1) && a.concurrent == 1)) return 0;
  return GPK_TUNE(GATE_KERNELS, 1) ? 2 : 1;
}

extern "C" int gpk_potrf(void* stream, double* A, int n, int extra, long lda, int batch,
                         long strideA, double* invd, int zero_upper, int* info) {
  return potrf_core((hipStream_t)stream, A, n, extra, lda, batch, strideA, invd, zero_upper, info);
}

extern "C" int gpk_potrf_inv(void* stream, double* A, int n, int extra, long lda, double* invd, int zero_upper,
                             int* info) {
  return potrf_core((hipStream_t)stream, A, n, extra + n, lda, 1, 0, invd, zero_upper, info, PotrfHooks(), n);
}

extern "C" int gpk_trtri_blocks(void* stream, const double* L, int n, long ldl, int batch,
                                long strideL, double* invd) {
  if (!L || !invd || n < 0) return GPK_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (batch <= 0) batch = 1;
  const int nblk = gpk_cdiv(n, NB);
  const long strideInv = (long)nblk * NB * NB;
  const int nfull = n / NB;
  for (int b = 0; b < batch; ++b) {
    double* Lb = const_cast<double*>(L) + (long)b * strideL;  // FACTORED leaf never writes A
    double* ib = invd + (long)b * strideInv;
    if (nfull > 0) {
      int rc = gpk_launch_leaf(s, Lb, ldl, (long)NB * (ldl + 1), NB, ib, (long)NB * NB, nullptr, 0,
                               nfull, 1);
      if (rc) return rc;
    }
    if (nfull < nblk) {
      const int j0 = nfull * NB;
      int rc = gpk_launch_leaf(s, Lb + (long)j0 * (ldl + 1), ldl, 0, n - j0, ib + (long)nfull * NB * NB,
                               0, nullptr, 0, 1, 1);
      if (rc) return rc;
    }
  }
  return 0;
}

// trans = 0:  B <- B L^-T  with (L, invd);   trans = 1:  B <- B L^-1 with (LT = L^T, invdT).
// Right-looking in column groups of 512: inside a group the 128-blocks are solved with their explicit inverses and
// each is followed by one K = 128 update of the rest of the group; then ONE K = 512 GEMM updates every column still
// to be solved.  (The left-looking form -- for every 128 columns a GEMM with 32 output tiles per 4096 rows and K up to
// n -- ran the N = 16384, T = 4096 predict solve at 7 TFLOP/s.)
extern "C" int gpk_trsm(void* stream, int trans, const double* L, long ldl, const double* invd,
                        int n, double* B, int m, long ldb, int batch, long strideL, long strideB) {
  if (!L || !invd || !B || n < 0 || m < 0) return GPK_E_ARG;
  if (n == 0 || m == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (batch <= 0) batch = 1;
  const long strideInv = (long)gpk_cdiv(n, NB) * NB * NB;
  int rc;
  if (trans == 0) {
    Bulk trsm_bulk;
    if (batch <= 1 && n >= 4096 && GPK_TUNE(TRSM_QUEUE, 0)) trsm_bulk.queue_cus = 256;   // (A/B, off: the cached-posterior solve 19.3 -> 20.2 ms with the queue)
    for (int g0 = 0; g0 < n; g0 += NBO) {
      rc = solve_group_fwd(s, trsm_bulk, B, ldb, B, ldb, m, L, ldl, invd, strideInv, n, g0, std::min(g0 + NBO, n), batch,
                           strideB, strideB, strideL);
      if (rc) return rc;
    }
  } else {
    const int ng = gpk_cdiv(n, NBO);
    for (int g = ng - 1; g >= 0; --g) {
      rc = solve_group_bwd(s, B, ldb, m, L, ldl, invd, strideInv, g * NBO, std::min((g + 1) * NBO, n), batch, strideB,
                           strideL);
      if (rc) return rc;
    }
  }
  return 0;
}

extern "C" int gpk_transpose_factor(void* stream, const double* L, long ldl, const double* invd,
                                    int n, double* LT, long ldlt, double* invdT) {
  if (!L || !invd || !LT || !invdT || n < 0) return GPK_E_ARG;
  if (n == 0) return 0;
  int rc = gpk_transpose(stream, L, n, n, ldl, LT, ldlt, 1, 1, 0, 0);
  if (rc) return rc;
  const int nblk = gpk_cdiv(n, NB);
  return gpk_transpose(stream, invd, NB, NB, NB, invdT, NB, 0, nblk, (long)NB * NB, (long)NB * NB);
}

// ---- projection:  ssq[p,b] = sum_j ( sum_k At[b,k] Lq_p[k,j] )^2 ---------------------------------------
extern "C" size_t gpk_project_workspace_bytes(int rows, int m, int P) {
  return (size_t)P * 2 * gpk_gemm_tiles_n(m) * rows * sizeof(double);
}

namespace {
// the GEMM alone: partials [P][nt = 2 * tiles_n][rows] in ws, one per 64 output columns
int project_parts(hipStream_t s, const double* At, int rows, int m, long ldat, long strideAt, const double* LqT, long ldl, int P, void* ws,
                  size_t ws_bytes) {
  if (!At || !LqT || rows < 0 || m <= 0 || P <= 0 || strideAt < 0) return GPK_E_ARG;
  if (!ws || ws_bytes < gpk_project_workspace_bytes(rows, m, P)) return GPK_E_WORKSPACE;
  if (rows == 0) return 0;
  const int nt = 2 * gpk_gemm_tiles_n(m);
  GemmArgs g = gemm_base(rows, m, m, 1.0, At, ldat, LqT, ldl, 0.0, nullptr, 0, P, strideAt, (long)m * ldl, 0);
  g.b_tri = 1;  // LqT[j,k] = Lq[k,j] vanishes for k < j
  g.epi = 1; g.sq_cols = m; g.c2_cols = 0;
  g.part = (double*)ws; g.part_ld = rows; g.stridePart = (long)nt * rows;
  g.C2 = (double*)ws; g.ldc2 = 0; g.strideC2 = 0;
  return gpk_launch_gemm(s, g);
}
}  // namespace

extern "C" int gpk_project_batched(void* stream, const double* At, int rows, int m, long ldat, long strideAt,
                                   const double* LqT, long ldl, int P, double* ssq, void* ws, size_t ws_bytes) {
  if (!ssq) return GPK_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int rc = project_parts(s, At, rows, m, ldat, strideAt, LqT, ldl, P, ws, ws_bytes);
  if (rc || rows == 0) return rc;
  const int nt = 2 * gpk_gemm_tiles_n(m);
  return gpk_launch_sum_parts(s, (const double*)ws, nt, rows, (long)nt * rows, P, ssq);
}

extern "C" int gpk_project(void* stream, const double* At, int rows, int m, long ldat,
                           const double* LqT, long ldl, int P, double* ssq, void* ws,
                           size_t ws_bytes) {
  return gpk_project_batched(stream, At, rows, m, ldat, 0, LqT, ldl, P, ssq, ws, ws_bytes);
}

// ---- fused driver: GPR.log_marginal_likelihood ----------------------------------------------------------
namespace {
struct LmlLayout {
  long ld; size_t off_T, off_invd, off_part, off_logdet, total;
};
LmlLayout lml_layout(int n, int P) {
  LmlLayout l{};
  l.ld = (long)gpk_align_up((size_t)n, 8);
  size_t o = 0;
  l.off_T = o; o += gpk_align_up((size_t)(n + P) * l.ld * sizeof(double), 256);
  l.off_invd = o; o += gpk_align_up(gpk_invd_elems(n, 1) * sizeof(double), 256);
  l.off_part = o; o += gpk_align_up((size_t)GPK_REDUCE_MAXPART * sizeof(double), 256);
  l.off_logdet = o; o += 256;
  l.total = o;
  return l;
}
}  // namespace

extern "C" size_t gpk_gpr_lml_workspace_bytes(int n, int d, int P) {
  (void)d;
  return lml_layout(n, P).total;
}

extern "C" int gpk_gpr_lml(void* stream, int family, const double* X, int n, int d, long ldx,
                           const double* Y, int P, long ldy, const double* ls_host, int ard,
                           double variance, double noise_variance, const double* noise_rows, double mean_const,
                           double* out, int* info, void* ws, size_t ws_bytes) {
  if (!X || !Y || !out || !info || n <= 0 || P <= 0) return GPK_E_ARG;
  const LmlLayout l = lml_layout(n, P);
  if (!ws || ws_bytes < l.total) return GPK_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  double* T = (double*)(w + l.off_T);
  double* invd = (double*)(w + l.off_invd);
  double* part = (double*)(w + l.off_part);
  double* logdet = (double*)(w + l.off_logdet);
  int rc;
  // K(X,X) + noise I, lower tiles only (gpr.py:100-101); a heteroskedastic likelihood (noise_rows: one variance per data
  // row, likelihoods/scalar_continuous.py:92-111) adds its vector to the diagonal instead (model_utils.py:46-50).
  // Large n (round 5 EXPERIMENT, off): the first outer panel's chain (5 leaves and their in-panel solves, ~0.5 ms) has nothing
  // to overlap with -- so only ITS columns are built before the factorisation starts and the remaining (n - w)^2 block is
  // built on the factorisation's bulk stream beside that chain (b_prologue).  Same formula per element: bit-identical
  // (tests at n = 4224 / 5000).  Measured at N = 16384: 31.13 - 31.21 against 31.17 - 31.23 ms (profiles/r05_ab_gpr_split_build.log):
  // the build's 65536 workgroups take every compute unit and the chain's one-workgroup leaves queue behind them, so the 0.4 ms
  // of build overlap ~0.05 ms of chain.  Kept behind GPK_GPR_SPLIT_BUILD in the A/B build.
  const int w0 = first_panel_columns(n);
  const bool split_build = n >= 4096 && w0 < n && GPK_TUNE(GPR_SPLIT_BUILD, 0);
  std::function<int(hipStream_t)> bpro;
  if (!split_build) {
    rc = gpk_kernel_matrix(stream, family, X, n, ldx, nullptr, 0, 0, d, ls_host, ard, variance,
                           noise_rows ? 0.0 : noise_variance, 1, T, l.ld);
    if (rc) return rc;
    if (noise_rows) {
      rc = gpk_diag_add(stream, T, n, l.ld, noise_rows);
      if (rc) return rc;
    }
  } else {
    // columns [0, w0): all rows (the w0 x w0 top block gets its upper triangle too; the factorisation never reads it)
    rc = gpk_kernel_matrix(stream, family, X, n, ldx, X, w0, ldx, d, ls_host, ard, variance, 0.0, 0, T, l.ld);
    if (rc) return rc;
    rc = noise_rows ? gpk_diag_add(stream, T, w0, l.ld, noise_rows) : gpk_launch_diag_add_scalar(s, T, w0, l.ld, noise_variance);
    if (rc) return rc;
    bpro = [&, w0](hipStream_t bs) -> int {
      double* Kb = T + (long)w0 * l.ld + w0;
      int r = gpk_kernel_matrix((void*)bs, family, X + (long)w0 * ldx, n - w0, ldx, nullptr, 0, 0, d, ls_host, ard, variance,
                                noise_rows ? 0.0 : noise_variance, 1, Kb, l.ld);
      if (r) return r;
      return noise_rows ? gpk_diag_add((void*)bs, Kb, n - w0, l.ld, noise_rows + w0) : 0;
    };
  }
  // (Y - m)^T as P extra rows (gpr.py:103, logdensities.py:149)
  rc = gpk_launch_transpose_shift(s, Y, n, P, ldy, T + (long)n * l.ld, l.ld, -mean_const);
  if (rc) return rc;
  // L = chol(K); extra rows -> alpha^T = (L^-1 (Y-m))^T  (gpr.py:102, logdensities.py:150)
  PotrfHooks hk;
  hk.b_prologue = split_build ? &bpro : nullptr;
  rc = potrf_core(s, T, n, P, l.ld, 1, 0, invd, 0, info, hk);
  if (rc) return rc;
  // p = -0.5 sum alpha^2 - 0.5 N log 2pi - sum log diag L, summed over the P columns
  rc = gpk_sum_log_diag(stream, T, n, l.ld, 1, 0, logdet);
  if (rc) return rc;
  int cnt = 0;
  rc = gpk_launch_sumsq_stage1(s, T + (long)n * l.ld, P, n, l.ld, 0, part, &cnt);
  if (rc) return rc;
  const double* parts[2] = {part, logdet};
  const int counts[2] = {cnt, 1};
  const double scales[2] = {-0.5, -(double)P};
  const double add = -0.5 * (double)n * (double)P * 1.8378770664093453;
  return gpk_launch_final(s, 2, parts, counts, scales, add, out);
}

// ---- fused driver: one shard of SVGP.elbo (whitened; shared kernel over the P latents) ----------------
namespace {
struct ElboLayout {
  long ld; int nt;
  size_t off_T, off_invd, off_LqT, off_s0, off_fmean, off_ssq, off_proj, off_part0, off_part1, off_part2, off_V, off_C, off_flags, off_Lfin, total;
};

ElboLayout elbo_layout(int m, int rows, int P, int q_diag, int whiten) {
  ElboLayout l{};
  l.ld = (long)gpk_align_up((size_t)m, 8);
  l.nt = 2 * gpk_gemm_tiles_n(m);
  size_t o = 0;
  // (minibatch rows padded to whole 32-row blocks: the single-launch step kernel runs full blocks only; the padding rows are
  // never initialised, never read by the multi-launch route and left out of the step kernel's final sum)
  const size_t rows_pad = gpk_align_up((size_t)rows, 32);
  // T [m + rows_pad rows], then -- directly behind it, so that the un-whitened form can use ONE trapezoid [Kuu ; Kfu ; q_mu^T ;
  // tril(q_sqrt_p)^T] with the minibatch rows unpadded -- room for P + P m more rows; the whitened form keeps its LqT there
  l.off_T = o; o += (size_t)(m + rows_pad) * l.ld * sizeof(double);
  // (un-whitened with a diagonal q_sqrt: the trapezoid is [Kuu ; Kfu ; q_mu^T ; I] -- P + m more rows)
  const size_t tail_rows = q_diag ? (whiten ? 0 : (size_t)P + m + 32) : (size_t)P + (size_t)P * m + 32;
  l.off_LqT = o; o = gpk_align_up(o + tail_rows * l.ld * sizeof(double), 256);
  l.off_invd = o; o += gpk_align_up(gpk_invd_elems(m, 1) * sizeof(double), 256);
  l.off_s0 = o; o += gpk_align_up((size_t)rows * sizeof(double), 256);
  l.off_fmean = o; o += gpk_align_up((size_t)rows * P * sizeof(double), 256);
  l.off_ssq = o; o += gpk_align_up((size_t)rows * P * sizeof(double), 256);
  // projection partials (full q_sqrt), or -- un-whitened with a diagonal q_sqrt -- the second solve A^T Lm^-1 [rows, ld]
  l.off_proj = o; o += q_diag ? (whiten ? 0 : gpk_align_up((size_t)rows * l.ld * sizeof(double), 256))
                              : gpk_align_up(gpk_project_workspace_bytes(rows, m, P), 256);
  l.off_part0 = o; o += gpk_align_up((size_t)GPK_REDUCE_MAXPART * sizeof(double), 256);
  l.off_part1 = o; o += gpk_align_up((size_t)GPK_REDUCE_MAXPART * sizeof(double), 256);
  l.off_part2 = o; o += gpk_align_up((size_t)(GPK_REDUCE_MAXPART + 64) * sizeof(double), 256);
  l.off_V = o; o += gpk_align_up((size_t)m * P * sizeof(double), 256);
  // single-launch step kernel (mega.hip; A/B build only, GPK_MEGA=1): projection accumulator [P, rows, ld], flag words and the
  // write-once copy of the factor.  The product library reserves nothing for it (round 4 did: +134 MB at Cm).
  l.off_C = l.off_flags = l.off_Lfin = o;
#ifdef GPK_EXPERIMENTAL
  if (!q_diag && GPK_TUNE(MEGA, GPK_MEGA_DEFAULT) && gpk_mega_supported(m, rows, P, 1 << 20)) {
    l.off_C = o; o += gpk_align_up((size_t)P * rows_pad * l.ld * sizeof(double), 256);
    l.off_flags = o; o += gpk_align_up(gpk_mega_flag_ints(m) * sizeof(int), 256);
    l.off_Lfin = o; o += gpk_align_up((size_t)m * l.ld * sizeof(double), 256);
  }
#endif
  l.total = o;
  return l;
}

int device_cus(int* ncu) {
  static int cached[16] = {0};
  int dev = 0;
  GPK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return GPK_E_UNSUPPORTED;
  if (!cached[dev]) {
    hipDeviceProp_t prop;
    GPK_HIP(hipGetDeviceProperties(&prop, dev));
    cached[dev] = prop.multiProcessorCount;
  }
  *ncu = cached[dev];
  return 0;
}
}  // namespace

#ifdef GPK_EXPERIMENTAL
// (A/B build only) byte offset of the step kernel's flag words / leaf time stamps inside the fused driver's workspace
extern "C" __attribute__((visibility("default"))) long gpk_exp_svgp_flags_offset(int m, int rows, int P) {
  return (long)elbo_layout(m, rows, P, 0, 1).off_flags;
}
#endif

extern "C" size_t gpk_svgp_elbo_workspace_bytes(int m, int rows, int d, int P, int q_diag, int whiten) {
  (void)d;
  return elbo_layout(m, rows, P, q_diag, whiten).total;
}

extern "C" int gpk_svgp_elbo_shard(void* stream, int family, const double* Z, int m, long ldz,
                                   const double* Xb, const double* Yb, int rows, long ldxb,
                                   long ldyb, int d, int P, const double* ls_host, int ard,
                                   double variance, double noise_variance, const double* noise_rows, double jitter,
                                   double mean_const, const double* q_mu, const double* q_sqrt,
                                   int q_diag, int whiten, double* out, int* info, void* ws,
                                   size_t ws_bytes) {
  if (!Z || !Xb || !Yb || !q_mu || !q_sqrt || !out || !info || m <= 0 || rows < 0 || P <= 0 || P > 16)
    return GPK_E_ARG;
  const ElboLayout l = elbo_layout(m, rows, P, q_diag, whiten);
  if (!ws || ws_bytes < l.total) return GPK_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  double* T = (double*)(w + l.off_T);
  if (!whiten && q_diag) {
    // ---- whiten = 0 with a DIAGONAL q_sqrt [m, P] (kullback_leiblers.py:128-165: diag branch with K; conditionals/util.py:139-149)
    // on ONE trapezoid [Kuu + jitter I ; Kfu ; q_mu^T ; I]: the identity rows come back as Lm^-T (written and solved by the
    // factorisation at m^3 / 3, gpk_potrf_inv's row skipping), which gives everything the reference takes from its two
    // factorisations and three triangular solves:
    //     A^T = Kfu Lm^-T (fvar's Knn - sum A^2),  a^T = (Lm^-1 q_mu)^T (Mahalanobis term),  (Kuu^-1)_ii = |row i of Lm^-T|^2 (trace term),
    //     A2^T = A^T Lm^-1 as one triangular-K GEMM (util.py:139's second solve of the minibatch columns) -> fmean = A2^T q_mu,
    //     ssq = sum_i (A2_ib q_sqrt_ip)^2 (util.py:149).
    double* invd_d = (double*)(w + l.off_invd);
    double* s0_d = (double*)(w + l.off_s0);
    double* fmean_d = (double*)(w + l.off_fmean);
    double* ssq_d = (double*)(w + l.off_ssq);
    double* pa = (double*)(w + l.off_part0);
    double* pb = (double*)(w + l.off_part1);
    double* pc = (double*)(w + l.off_part2);          // [MAXPART] trace / log det q partials, then [1] log det Lm
    double* Kfu_d = T + (long)m * l.ld;
    double* arow = Kfu_d + (long)rows * l.ld;          // [P, m]
    double* LinvT = arow + (long)P * l.ld;             // [m, ld]: Lm^-T (upper triangular)
    double* A2 = (double*)(w + l.off_proj);            // [rows, ld]
    int rcd = 0;
    const std::function<int(hipStream_t)> kuu_d = [&](hipStream_t ps) -> int {
      return gpk_kernel_matrix((void*)ps, family, Z, m, ldz, nullptr, 0, 0, d, ls_host, ard, variance, jitter, 1, T, l.ld);
    };
    const std::function<int(hipStream_t)> prod = [&](hipStream_t xs) -> int {
      int r = gpk_kernel_matrix((void*)xs, family, Xb, rows, ldxb, Z, m, ldz, d, ls_host, ard, variance, 0.0, 0, Kfu_d, l.ld);
      if (r) return r;
      return gpk_transpose((void*)xs, q_mu, m, P, P, arow, l.ld, 0, 1, 0, 0);
    };
    PotrfHooks hkd;
    hkd.x_prologue = &prod;
    hkd.p_prologue = &kuu_d;
    rcd = potrf_core(s, T, m, rows + P + m, l.ld, 1, 0, invd_d, 0, info, hkd, m);
    if (rcd) return rcd;
    if (rows > 0) {
      GemmArgs g = gemm_base(rows, m, m, 1.0, Kfu_d, l.ld, LinvT, l.ld, 0.0, A2, l.ld, 1, 0, 0, 0);
      g.b_tri = 1;  // LinvT[j, k] = Lm^-1[k, j] vanishes for k < j
      rcd = gpk_launch_gemm(s, g);
      if (rcd) return rcd;
      rcd = gpk_row_sumsq(stream, Kfu_d, rows, m, l.ld, 1.0, 0.0, s0_d);
      if (rcd) return rcd;
      rcd = gpk_row_stats(stream, A2, rows, m, l.ld, q_mu, q_sqrt, P, 1.0, 0.0, nullptr, fmean_d, ssq_d);
      if (rcd) return rcd;
    }
    int ca = 0;
    rcd = gpk_launch_varexp_stage1(s, Yb, ldyb, fmean_d, rows, P, s0_d, 0, ssq_d, &variance, 0, noise_variance, mean_const, nullptr,
                                   pa, &ca, noise_rows);
    if (rcd) return rcd;
    const double* q0[1] = {pa};
    const double one_d = 1.0;
    rcd = gpk_launch_final(s, 1, q0, &ca, &one_d, 0.0, out);
    if (rcd) return rcd;
    // KL = 0.5 ( |a|^2 + sum_i [(Kuu^-1)_ii sum_p w_ip^2 - sum_p log w_ip^2] - M P ) + P sum log diag(Lm)
    int cm_ = 0, ct = 0;
    rcd = gpk_launch_sumsq_stage1(s, arow, P, m, l.ld, 0, pb, &cm_);
    if (rcd) return rcd;
    rcd = gpk_launch_kl_unwhite_diag_stage1(s, LinvT, l.ld, m, q_sqrt, P, pc, &ct);
    if (rcd) return rcd;
    double* ldl = pc + GPK_REDUCE_MAXPART;
    rcd = gpk_sum_log_diag(stream, T, m, l.ld, 1, 0, ldl);
    if (rcd) return rcd;
    const double* kp[3] = {pb, pc, ldl};
    const int kc[3] = {cm_, ct, 1};
    const double ks[3] = {0.5, 0.5, (double)P};
    return gpk_launch_final(s, 3, kp, kc, ks, -0.5 * (double)m * (double)P, out + 1);
  }
  if (!whiten) {
    // ---- whiten = 0 (kullback_leiblers.py:98-165 with K = Kuu, conditionals/util.py:128-167 with white = False) on ONE
    // trapezoid [Kuu + jitter I ; Kfu ; q_mu^T ; tril(q_sqrt_p)^T].  The reference factors Kuu twice (once for the KL, once for
    // the conditional) and solves the minibatch columns twice (Lm^-1, then Lm^-T).  Here the extra rows come back as
    //     A^T = Kfu Lm^-T,   a^T = (Lm^-1 q_mu)^T,   G_p^T = (Lm^-1 Lq_p)^T   (G_p lower triangular again)
    // which are the Mahalanobis / trace terms of the KL AND the whitened parameters of the same q(u): fmean = A^T a,
    // sum_j (Lq^T Lm^-T A)_j^2 = sum_j (G^T A)_j^2 -- the projection kernel of the whitened path with G^T in place of Lq^T,
    // no second triangular solve of the minibatch rows.
    double* invd_u = (double*)(w + l.off_invd);
    double* s0_u = (double*)(w + l.off_s0);
    double* fmean_u = (double*)(w + l.off_fmean);
    double* ssq_u = (double*)(w + l.off_ssq);
    double* pa = (double*)(w + l.off_part0);
    double* pb = (double*)(w + l.off_part1);
    double* pc = (double*)(w + l.off_part2);          // [MAXPART] trace partials, then [P] log det q, then [1] log det Lm
    double* V = (double*)(w + l.off_V);
    double* Kfu_u = T + (long)m * l.ld;
    double* arow = Kfu_u + (long)rows * l.ld;          // [P, m]
    double* GT = arow + (long)P * l.ld;                // [P][m][ld]
    int rcu = 0;
    const std::function<int(hipStream_t)> kuu_u = [&](hipStream_t ps) -> int {
      return gpk_kernel_matrix((void*)ps, family, Z, m, ldz, nullptr, 0, 0, d, ls_host, ard, variance, jitter, 1, T, l.ld);
    };
    const std::function<int(hipStream_t)> pro = [&](hipStream_t xs) -> int {
      int r = gpk_kernel_matrix((void*)xs, family, Xb, rows, ldxb, Z, m, ldz, d, ls_host, ard, variance, 0.0, 0, Kfu_u, l.ld);
      if (r) return r;
      r = gpk_transpose((void*)xs, q_mu, m, P, P, arow, l.ld, 0, 1, 0, 0);
      if (r) return r;
      return gpk_transpose((void*)xs, q_sqrt, m, m, m, GT, l.ld, 1, P, (long)m * m, (long)m * l.ld);
    };
    // (P = 1: the m rows of tril(q_sqrt)^T are the LAST rows of the trapezoid and upper triangular -- row j stays zero left of
    //  column j until its column group is reached, so the row solve skips them there: 3/8 of their work, round 5)
    PotrfHooks hku;
    hku.x_prologue = &pro;
    hku.p_prologue = &kuu_u;
    rcu = potrf_core(s, T, m, rows + P + P * m, l.ld, 1, 0, invd_u, 0, info, hku, P == 1 ? m : 0, true);
    if (rcu) return rcu;
    rcu = gpk_transpose(stream, arow, P, m, l.ld, V, P, 0, 1, 0, 0);           // a = Lm^-1 q_mu as [m, P]
    if (rcu) return rcu;
    rcu = gpk_row_stats(stream, Kfu_u, rows, m, l.ld, V, nullptr, P, 1.0, 0.0, s0_u, fmean_u, nullptr);
    if (rcu) return rcu;
    rcu = gpk_project(stream, Kfu_u, rows, m, l.ld, GT, l.ld, P, ssq_u, w + l.off_proj, gpk_project_workspace_bytes(rows, m, P));
    if (rcu) return rcu;
    int ca = 0;
    rcu = gpk_launch_varexp_stage1(s, Yb, ldyb, fmean_u, rows, P, s0_u, 0, ssq_u, &variance, 0, noise_variance, mean_const, nullptr,
                                   pa, &ca, noise_rows);
    if (rcu) return rcu;
    const double* q0[1] = {pa};
    const double one_u = 1.0;
    rcu = gpk_launch_final(s, 1, q0, &ca, &one_u, 0.0, out);
    if (rcu) return rcu;
    // KL = 0.5 |a|^2 + 0.5 sum_p |G_p|_F^2 - 0.5 M P - 0.5 sum log diag(Lq)^2 + P sum log diag(Lm)
    int cm_ = 0, ct = 0;
    rcu = gpk_launch_sumsq_stage1(s, arow, P, m, l.ld, 0, pb, &cm_);
    if (rcu) return rcu;
    rcu = gpk_launch_sumsq_stage1(s, GT, P * m, m, l.ld, 0, pc, &ct);
    if (rcu) return rcu;
    double* ldq = pc + GPK_REDUCE_MAXPART;
    double* ldl = ldq + P;
    rcu = gpk_launch_sum_log_diag_sq(s, q_sqrt, m, m, P, (long)m * m, ldq);
    if (rcu) return rcu;
    rcu = gpk_sum_log_diag(stream, T, m, l.ld, 1, 0, ldl);
    if (rcu) return rcu;
    const double* kp[4] = {pb, pc, ldq, ldl};
    const int kc[4] = {cm_, ct, P, 1};
    const double ks[4] = {0.5, 0.5, -0.5, (double)P};
    return gpk_launch_final(s, 4, kp, kc, ks, -0.5 * (double)m * (double)P, out + 1);
  }
  double* invd = (double*)(w + l.off_invd);
  double* LqT = (double*)(w + l.off_LqT) + (q_diag ? 0 : (long)P * l.ld);   // (behind the P rows the un-whitened form keeps there)
  double* s0 = (double*)(w + l.off_s0);
  double* fmean = (double*)(w + l.off_fmean);
  double* ssq = (double*)(w + l.off_ssq);
  double* part0 = (double*)(w + l.off_part0);
  double* part1 = (double*)(w + l.off_part1);
  double* Kfu = T + (long)m * l.ld;  // extra rows of the trapezoid: Kfu in, A^T = Kfu Lm^-T out (in place)
  int rc;
  // Kuf^T = k(Xb, Z) as the extra rows (posteriors.py:836, covariances/kufs.py:31-34).  Only the bulk stream of the
  // factorisation consumes it, so it is built THERE (ordered after everything already queued on the caller's stream)
  // and the panel chain starts right after the much smaller Kuu build.  Work that depends on neither factorisation
  // nor minibatch solve -- tril(q_sqrt)^T for the projection and the whole KL term -- goes to that stream too, which
  // idles until the first 512 columns of Lm exist; gpk_potrf joins it.
#ifdef GPK_EXPERIMENTAL
  // ---- single-launch route (mega.hip): builds + KL on the caller's stream, then ONE persistent kernel for everything
  // that depends on the factorisation.  Taken when the shapes fit one row block per compute unit.
  if (!q_diag && !noise_rows && GPK_TUNE(MEGA, GPK_MEGA_DEFAULT) && rows > 0) {
    int ncu = 0;
    rc = device_cus(&ncu);
    if (rc) return rc;
    if (gpk_mega_supported(m, rows, P, ncu)) {
      rc = gpk_kernel_matrix(stream, family, Z, m, ldz, nullptr, 0, 0, d, ls_host, ard, variance, jitter, 1, T, l.ld);
      if (rc) return rc;
      rc = gpk_kernel_matrix(stream, family, Xb, rows, ldxb, Z, m, ldz, d, ls_host, ard, variance, 0.0, 0, Kfu, l.ld);
      if (rc) return rc;
      rc = gpk_transpose(stream, q_sqrt, m, m, m, LqT, l.ld, 1, P, (long)m * m, (long)m * l.ld);
      if (rc) return rc;
      int ck = 0;
      rc = gpk_launch_kl_white_stage1(s, q_mu, q_sqrt, m, P, q_diag, part1, &ck);
      if (rc) return rc;
      const double* pk[1] = {part1};
      const double halfk = 0.5;
      rc = gpk_launch_final(s, 1, pk, &ck, &halfk, -0.5 * (double)m * (double)P, out + 1);
      if (rc) return rc;
      GPK_HIP(hipMemsetAsync(info, 0, sizeof(int), s));
      return gpk_launch_svgp_mega(s, GPK_TUNE(MEGA_PROTO, 1), ncu, T, l.ld, m, rows, invd, (double*)(w + l.off_Lfin), LqT, l.ld, (double*)(w + l.off_C), q_mu, P,
                                  Yb, ldyb, s0, fmean, ssq, part0, (int*)(w + l.off_flags), info, out, variance, noise_variance,
                                  mean_const, GPK_TUNE(MEGA_MIN_WGS, 96));
    }
  }
#endif
  const bool side = m > GPK_NB && m < 4096 && rows > 256;
  // Kuu + jitter I (posteriors.py:835, covariances/kuus.py:29-34), lower tiles only: the chain's first leaf waits for
  // nothing else, so the factorisation enqueues it on its panel stream, directly in front of that leaf
  const std::function<int(hipStream_t)> kuu_build = [&](hipStream_t ps) -> int {
    return gpk_kernel_matrix((void*)ps, family, Z, m, ldz, nullptr, 0, 0, d, ls_host, ard, variance, jitter, 1, T, l.ld);
  };
  int c1 = 0;
  // everything else that precedes the minibatch solve, as one closure: enqueued by the factorisation on its bulk stream
  // (side) or here on the caller's stream
  const std::function<int(hipStream_t)> prologue = [&](hipStream_t xs) -> int {
    return gpk_kernel_matrix((void*)xs, family, Xb, rows, ldxb, Z, m, ldz, d, ls_host, ard, variance, 0.0, 0, Kfu, l.ld);
  };
  // tril(q_sqrt)^T for the projection and the whole KL term depend on neither the factorisation nor the minibatch solve
  const std::function<int(hipStream_t)> late = [&](hipStream_t xs) -> int {
    int r = 0;
    if (!q_diag) {
      r = gpk_transpose((void*)xs, q_sqrt, m, m, m, LqT, l.ld, 1, P, (long)m * m, (long)m * l.ld);
      if (r) return r;
    }
    r = gpk_launch_kl_white_stage1(xs, q_mu, q_sqrt, m, P, q_diag, part1, &c1);
    if (r) return r;
    const double* p1s[1] = {part1};
    const double halfs = 0.5;
    return gpk_launch_final(xs, 1, p1s, &c1, &halfs, -0.5 * (double)m * (double)P, out + 1);
  };
  // Lm = chol(Kuu);  A^T = Kfu Lm^-T   (conditionals/util.py:67,125)
  PotrfHooks hk;
  hk.x_prologue = &prologue;
  hk.p_prologue = &kuu_build;
  hk.late_work = side ? &late : nullptr;
  // s0 = sum_k A^2 (util.py:133), fmean = A^T q_mu (util.py:144), q_diag: ssq = sum (A q_sqrt)^2 (:149)
  const std::function<int(hipStream_t)> stats = [&](hipStream_t xs) -> int {
    return gpk_row_stats((void*)xs, Kfu, rows, m, l.ld, q_mu, q_diag ? q_sqrt : nullptr, P, 1.0, 0.0, s0, fmean, q_diag ? ssq : nullptr);
  };
  XTail xt;
  if (!q_diag && rows > 0) {   // (beside the projection GEMM; with a diagonal q_sqrt nothing would run beside it)
    xt.work = &stats;
    hk.x_tail = &xt;
  }
  rc = potrf_core(s, T, m, rows, l.ld, 1, 0, invd, 0, info, hk);
  if (rc) return rc;
  if (!xt.used) {
    rc = stats(s);
    if (rc) return rc;
  }
  VarexpExtra ex;
  if (xt.used) { ex.wait_ptr = xt.done_ptr; ex.wait_val = xt.done_val; ex.wait_info = info; }
  if (!q_diag) {
    // L = band_part(q_sqrt,-1,0); LTA = L^T A; ssq = sum LTA^2   (util.py:151-164)
    if (!side) {
      rc = gpk_transpose(stream, q_sqrt, m, m, m, LqT, l.ld, 1, P, (long)m * m, (long)m * l.ld);
      if (rc) return rc;
    }
    // (the column-slot partials of the projection are summed by the variational-expectation kernel: no sum_parts launch)
    rc = project_parts(s, Kfu, rows, m, l.ld, 0, LqT, l.ld, P, w + l.off_proj, gpk_project_workspace_bytes(rows, m, P));
    if (rc) return rc;
    if (rows > 0 && GPK_TUNE(VAREXP_SUMS_PARTS, 0)) {
      ex.ssq_part = (const double*)(w + l.off_proj);
      ex.ssq_nt = 2 * gpk_gemm_tiles_n(m);
      ex.ssq_stride = (long)ex.ssq_nt * rows;
    } else if (rows > 0) {
      rc = gpk_launch_sum_parts(s, (const double*)(w + l.off_proj), 2 * gpk_gemm_tiles_n(m), rows, (long)2 * gpk_gemm_tiles_n(m) * rows, P, ssq);
      if (rc) return rc;
    }
  }
  // sum_b var_exp_b  (likelihoods/scalar_continuous.py:139-148, svgp.py:174,181)
  int c0 = 0;
  rc = gpk_launch_varexp_stage1(s, Yb, ldyb, fmean, rows, P, s0, 0, ssq, &variance, 0, noise_variance,
                                mean_const, nullptr, part0, &c0, noise_rows, &ex);
  if (rc) return rc;
  const double* p0[1] = {part0};
  const double one = 1.0;
  rc = gpk_launch_final(s, 1, p0, &c0, &one, 0.0, out);
  if (rc) return rc;
  if (side) return 0;
  // KL[q || N(0, I)]  (kullback_leiblers.py:45-46, 98-165)
  rc = gpk_launch_kl_white_stage1(s, q_mu, q_sqrt, m, P, q_diag, part1, &c1);
  if (rc) return rc;
  const double* p1[1] = {part1};
  const double half = 0.5;
  return gpk_launch_final(s, 1, p1, &c1, &half, -0.5 * (double)m * (double)P, out + 1);
}

// ---- fused driver: one shard of SVGP.elbo with SEPARATE kernels per latent (SeparateIndependent, whitened, full q_sqrt) --------
namespace {
struct ElboSepLayout {
  long ld, strideT;
  size_t off_T, off_invd, off_LqT, off_s0, off_fmean, off_ssq, off_proj, off_part0, off_part1, total;
};
ElboSepLayout elbo_sep_layout(int m, int rows, int P) {
  ElboSepLayout l{};
  l.ld = (long)gpk_align_up((size_t)m, 8);
  l.strideT = (long)(m + rows) * l.ld;
  size_t o = 0;
  l.off_T = o; o += gpk_align_up((size_t)P * l.strideT * sizeof(double), 256);
  l.off_invd = o; o += gpk_align_up(gpk_invd_elems(m, P) * sizeof(double), 256);
  l.off_LqT = o; o += gpk_align_up((size_t)P * m * l.ld * sizeof(double), 256);
  l.off_s0 = o; o += gpk_align_up((size_t)rows * P * sizeof(double), 256);
  l.off_fmean = o; o += gpk_align_up((size_t)rows * P * sizeof(double), 256);
  l.off_ssq = o; o += gpk_align_up((size_t)rows * P * sizeof(double), 256);
  l.off_proj = o; o += gpk_align_up(gpk_project_workspace_bytes(rows, m, P), 256);
  l.off_part0 = o; o += gpk_align_up((size_t)GPK_REDUCE_MAXPART * sizeof(double), 256);
  l.off_part1 = o; o += gpk_align_up((size_t)GPK_REDUCE_MAXPART * sizeof(double), 256);
  l.total = o;
  return l;
}
}  // namespace

extern "C" size_t gpk_svgp_elbo_sep_workspace_bytes(int m, int rows, int d, int P) {
  (void)d;
  return elbo_sep_layout(m, rows, P).total;
}

// The P problems of conditionals/util.py:566-629 (tf.map_fn over the latents) share nothing but the minibatch: P covariance
// pairs built straight into ONE batched trapezoid [P][(m + rows) x ld], one batched factorisation with the minibatch rows riding
// along (gpk_potrf, batch = P), one batched row-statistics launch, one batched projection, one reduction.  Composed from the
// Python mirror the same step issues ~50 launches with host gaps between them (profiles/r04_c5sep_timeline_composed.txt).
// (Measured and not kept: the extra rows solved out of place against EXPLICIT 512-column group inverses -- nine short launches
// for the inverses + one triangular-K GEMM per group instead of the fused in-group kernel: 2.15 / 2.16 against 2.14 ms.)
extern "C" int gpk_svgp_elbo_shard_sep(void* stream, const int* family_host, const double* Z, int m, long ldz, long strideZ,
                                       const double* Xb, const double* Yb, int rows, long ldxb, long ldyb, int d, int P,
                                       const double* ls_host, int ard, const double* variance_host, double noise_variance,
                                       const double* noise_rows, double jitter, double mean_const, const double* q_mu,
                                       const double* q_sqrt, double* out,
                                       int* info, void* ws, size_t ws_bytes) {
  if (!family_host || !Z || !Xb || !Yb || !q_mu || !q_sqrt || !ls_host || !variance_host || !out || !info || m <= 0 || rows < 0 ||
      P <= 0 || P > 16 || d <= 0 || strideZ < 0)
    return GPK_E_ARG;
  const ElboSepLayout l = elbo_sep_layout(m, rows, P);
  if (!ws || ws_bytes < l.total) return GPK_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  double* T = (double*)(w + l.off_T);
  double* invd = (double*)(w + l.off_invd);
  double* LqT = (double*)(w + l.off_LqT);
  double* s0 = (double*)(w + l.off_s0);
  double* fmean = (double*)(w + l.off_fmean);
  double* ssq = (double*)(w + l.off_ssq);
  double* part0 = (double*)(w + l.off_part0);
  double* part1 = (double*)(w + l.off_part1);
  const int nls = ard ? d : 1;
  int rc;
  // Kuu_p + jitter I (lower tiles): the chain's first (batched) leaf waits for nothing else -- enqueued on the panel stream
  const std::function<int(hipStream_t)> kuu_build = [&](hipStream_t ps) -> int {
    for (int p = 0; p < P; ++p) {
      const int r = gpk_kernel_matrix((void*)ps, family_host[p], Z + (long)p * strideZ, m, ldz, nullptr, 0, 0, d, ls_host + (long)p * nls,
                                      ard, variance_host[p], jitter, 1, T + (long)p * l.strideT, l.ld);
      if (r) return r;
    }
    return 0;
  };
  const bool side = m > GPK_NB && m < 4096 && rows > 256;
  int c1 = 0;
  auto kl_and_transpose = [&](hipStream_t xs) -> int {
    int r = gpk_transpose((void*)xs, q_sqrt, m, m, m, LqT, l.ld, 1, P, (long)m * m, (long)m * l.ld);
    if (r) return r;
    r = gpk_launch_kl_white_stage1(xs, q_mu, q_sqrt, m, P, 0, part1, &c1);
    if (r) return r;
    const double* p1s[1] = {part1};
    const double halfs = 0.5;
    return gpk_launch_final(xs, 1, p1s, &c1, &halfs, -0.5 * (double)m * (double)P, out + 1);
  };
  const std::function<int(hipStream_t)> prologue = [&](hipStream_t xs) -> int {
    for (int p = 0; p < P; ++p) {
      const int r = gpk_kernel_matrix((void*)xs, family_host[p], Xb, rows, ldxb, Z + (long)p * strideZ, m, ldz, d,
                                      ls_host + (long)p * nls, ard, variance_host[p], 0.0, 0, T + (long)p * l.strideT + (long)m * l.ld,
                                      l.ld);
      if (r) return r;
    }
    return 0;
  };
  const std::function<int(hipStream_t)> late = [&](hipStream_t xs) -> int { return kl_and_transpose(xs); };
  PotrfHooks hk;
  hk.x_prologue = &prologue;
  hk.p_prologue = &kuu_build;
  hk.late_work = side ? &late : nullptr;
  rc = potrf_core(s, T, m, rows, l.ld, P, l.strideT, invd, 0, info, hk);
  if (rc) return rc;
  if (!side) {
    rc = kl_and_transpose(s);
    if (rc) return rc;
  }
  const double* At = T + (long)m * l.ld;
  rc = gpk_launch_row_stats_sep(s, At, l.strideT, rows, m, l.ld, q_mu, P, s0, fmean);
  if (rc) return rc;
  rc = gpk_project_batched(stream, At, rows, m, l.ld, l.strideT, LqT, l.ld, P, ssq, w + l.off_proj,
                           gpk_project_workspace_bytes(rows, m, P));
  if (rc) return rc;
  int c0 = 0;
  rc = gpk_launch_varexp_stage1(s, Yb, ldyb, fmean, rows, P, s0, 1, ssq, variance_host, 1, noise_variance, mean_const, nullptr, part0,
                                &c0, noise_rows);
  if (rc) return rc;
  const double* p0[1] = {part0};
  const double one = 1.0;
  return gpk_launch_final(s, 1, p0, &c0, &one, 0.0, out);
}
