// Host-side orchestration (no device code here): the trapezoidal blocked Cholesky, triangular solves
// against a cached factor, the projection onto q_sqrt, and the two fused model drivers
// (GPR.log_marginal_likelihood, one shard of SVGP.elbo).
//
// Trapezoidal Cholesky.  A is [(n + extra) x n]: the top square block is factored, the `extra` rows
// below ride along through every panel solve and trailing update and come out as  B L^-T  -- the
// tf.linalg.triangular_solve of the reference fused into the factorisation.  Two-level right-looking:
//   outer panels of 768 columns (n >= 4096) -> trailing update is a K = 768 MFMA GEMM,
//   inner blocks of NB = 128 columns        -> leaf kernel (L11 and L11^-1), in-place panel solve
//                                              A21 <- A21 * L11^-T as a GEMM, update of the rest of the panel.
// For n < 4096 (the SVGP sizes) the outer panel IS one 128-column block: the factorisation is a latency chain
// leaf -> panel solve -> strip, and everything that is not on that chain (the solve of the minibatch rows, the
// projection onto q_sqrt) runs beside it as bulk work under a SOFTWARE CU reservation (gemm.hip, ticketed tiles).
#include "gpk_internal.h"
#include <algorithm>
#include <functional>
#include <map>
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

// how a bulk GEMM is launched: plain, or ticketed with the software CU reservation (batch 1 only)
struct Bulk {
  unsigned* ctr = nullptr;
  const unsigned char* resv = nullptr;
  int cap = 0;    // plain launches only: cap on persistent workgroups of big updates (A/B of the round-1 scheme)
  int min_k = 0;  // ticketing only for launches with K >= min_k: the long ones, which would otherwise hold every CU for
                  // hundreds of microseconds; the short in-group launches (K = 128) stay one-shot kernels
  void apply(GemmArgs& g) const {
    if (!ctr && cap > 0 && g.k >= 256) g.max_wgs = cap;
    if (ctr && g.batch == 1 && g.k >= min_k) {
      g.ctr = ctr;
      g.resv = resv;
      g.no_small = 1;  // the one-shot LDS-DMA kernel has no reservation check: everything goes through the tiled kernel
    }
  }
};
}  // namespace

extern "C" const char* gpk_version(void) {
#ifdef GPK_EXPERIMENTAL
  return "gpk 0.2 (gfx950, fp64 MFMA) [A/B build: environment tunables enabled]";
#else
  return "gpk 0.2 (gfx950, fp64 MFMA)";
#endif
}

extern "C" size_t gpk_invd_elems(int n, int batch) {
  return (size_t)(batch > 0 ? batch : 1) * gpk_cdiv(n, NB) * NB * NB;
}

// ---- per-device internal state (created lazily, once; see gpk.h "Internal state and threading") -------------------
// The factorisation runs on streams of its own, forked from / joined to the caller's stream with events only:
//   P   "panel" stream, high priority: the latency-bound critical path (leaf, panel solve, inner updates, strip) of
//       the NEXT outer panel (look-ahead);
//   B   bulk stream of LARGE factorisations (n >= 4096): the big MFMA GEMMs of the outer trailing updates and of the
//       extra rows.  CU-masked in hardware: its mask leaves 16 compute units (2 per XCD; mask bit i is CU i/8 of XCD
//       i%8 on MI355X, tools/cumask_test.hip) to the panel stream -- without that the one-workgroup leaf kernel, which
//       needs a whole CU's LDS, queues behind thousands of resident GEMM workgroups (a 2 ms stall per panel at
//       N = 16384) and the look-ahead never overlaps;
//   Bl  the same for the chain-bound tail of a large factorisation: leaves half the CUs to P;
//   Bs  rest-updates of SMALL factorisations (they are ON the critical path there): all CUs;
//   X   bulk stream of small factorisations: the right-looking solve of the extra rows (the SVGP minibatch) and the
//       streamed projection.  Unmasked (CU-masked queues dispatched these short kernels slowly: -20 %); its GEMMs are
//       ticketed and honour the software reservation table `resv` instead (gemm.hip).
// One std::recursive_mutex per device serialises the ENQUEUE of factorisations (shared streams, event pool, ticket
// counters); the enqueued work of successive calls is ordered by the streams themselves.
namespace {
struct Aux {
  std::recursive_mutex mu;
  bool ready = false;
  hipStream_t P = nullptr, B = nullptr, Bs = nullptr, Bl = nullptr, X = nullptr, pad = nullptr;
  hipEvent_t* ev = nullptr;
  int nev = 0;
  unsigned char* resv = nullptr;  // device [GPK_CU_KEYS]: 1 = compute unit reserved for the latency chain
  unsigned* ctr = nullptr;        // device [16]: ticket counters of stream X (self-resetting)
  int ncu = 0, bulk_cus = 0, resv_cus = 0;
  // tile-dataflow bulk kernel: state words [16 tickets | FLOW_MAX_RB progress counters] (zeroed before every launch),
  // chain flags (never reset: they carry the epoch of the factorisation that raised them), cached task lists per shape
  unsigned* flow_state = nullptr;
  unsigned* flow_flags = nullptr;
  unsigned epoch = 0;
  // chain-panel schedule (A/B build): bulk streams masked away from the CUs the chain launches need, flag words
  hipStream_t Mb = nullptr, Mr = nullptr, Mx = nullptr, Mpad = nullptr;
  unsigned* ck_flagL = nullptr; unsigned long long* ck_cnt = nullptr;  // device memory
  unsigned* ck_flagK = nullptr; unsigned* ck_flagB = nullptr;          // signal memory (stream value waits / writes)
  unsigned ck_epoch = 0;
  struct Plan { int n = 0, rows = 0, P = 0, proj = 0; FlowTask* dev = nullptr; int off[9] = {0}; std::vector<FlowGroup> groups; };
  Plan plans[4];
  int nplans = 0;
};
constexpr int FLOW_MAX_RB = 1024;  // row blocks of 128 rows: minibatches up to 131072 rows
Aux g_aux[16];

int masked_stream(hipStream_t* out, int ncu, int first, int last) {  // CUs [first, last)
  if (first <= 0 && last >= ncu) return (int)hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  uint32_t mask[32] = {0};
  for (int i = first; i < last; ++i) mask[i >> 5] |= 1u << (i & 31);
  return (int)hipExtStreamCreateWithCUMask(out, (uint32_t)((ncu + 31) / 32), mask);
}

// Census of the physical CUs (key = XCC, SE, SH, CU) and choice of `want` of them, the same number on every XCD and
// spread over its shader engines (highest CU ids first), as the software reservation table.
int build_reservation(Aux& a, int want) {
  const int n = 1024;
  unsigned* keys_dev = nullptr;
  GPK_HIP(hipMalloc((void**)&keys_dev, sizeof(unsigned) * n));
  int rc = gpk_cu_census(a.P, keys_dev, n);
  std::vector<unsigned> keys(n);
  if (rc == 0) rc = (int)hipMemcpyAsync(keys.data(), keys_dev, sizeof(unsigned) * n, hipMemcpyDeviceToHost, a.P);
  if (rc == 0) rc = (int)hipStreamSynchronize(a.P);
  (void)hipFree(keys_dev);
  if (rc) return rc;
  std::map<unsigned, std::map<unsigned, std::vector<unsigned>>> chip;  // xcc -> se -> cu keys
  for (unsigned k : keys) {
    k &= GPK_CU_KEYS - 1;
    auto& v = chip[k >> 8][(k >> 5) & 7];
    if (std::find(v.begin(), v.end(), k) == v.end()) v.push_back(k);
  }
  std::vector<unsigned char> table(GPK_CU_KEYS, 0);
  int chosen = 0;
  const int per_xcc = chip.empty() ? 0 : want / (int)chip.size();
  for (auto& xcc : chip) {
    for (auto& se : xcc.second) std::sort(se.second.begin(), se.second.end());
    int got = 0;
    while (got < per_xcc) {
      bool any = false;
      for (auto& se : xcc.second) {
        if (got >= per_xcc) break;
        if (se.second.size() <= 1) continue;  // never take the last CU of a shader engine
        table[se.second.back()] = 1;
        se.second.pop_back();
        ++got; ++chosen;
        any = true;
      }
      if (!any) break;
    }
  }
  GPK_HIP(hipMalloc((void**)&a.resv, GPK_CU_KEYS));
  GPK_HIP(hipMemcpy(a.resv, table.data(), GPK_CU_KEYS, hipMemcpyHostToDevice));
  GPK_HIP(hipMalloc((void**)&a.ctr, sizeof(unsigned) * 16));
  GPK_HIP(hipMemset(a.ctr, 0, sizeof(unsigned) * 16));
  a.resv_cus = chosen;
  GPK_TRACE("reservation: %zu XCDs seen, %d compute units reserved\n", chip.size(), chosen);
  return 0;
}

// (caller holds a.mu)
int aux_get(int dev, int need, Aux** out) {
  Aux& a = g_aux[dev];
  if (!a.ready) {
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
    //   and 5 (pipes 3 and 0);  then the masked B -> pipe 3 (or 1) and Bl -> pipe 0 (or 2).
    // The chain (P), its rest-updates (Bs) and the bulk stream (X or B) are then always on three different pipes.
    GPK_HIP(hipStreamCreateWithPriority(&a.P, hipStreamNonBlocking, hi));
    GPK_HIP(hipStreamCreateWithFlags(&a.X, hipStreamNonBlocking));
    GPK_HIP(hipStreamCreateWithFlags(&a.pad, hipStreamNonBlocking));
    GPK_HIP(hipStreamCreateWithFlags(&a.Bs, hipStreamNonBlocking));
    int reserved = GPK_TUNE(RESERVED_CUS, 8);
    if (ncu > 1024 || reserved < 0 || reserved >= ncu) reserved = 0;
    int rc = masked_stream(&a.B, ncu, reserved, ncu);
    if (rc) return rc;
    a.bulk_cus = ncu - reserved;
    int late_res = GPK_TUNE(LATE_RESERVED_CUS, -1);
    if (late_res < 0) late_res = ncu / 2;
    if (ncu > 1024 || late_res >= ncu) late_res = 0;
    rc = masked_stream(&a.Bl, ncu, late_res, ncu);
    if (rc) return rc;
    if (kGpkExp) {  // the software CU reservation is an A/B-build experiment (gpk_internal.h)
      rc = build_reservation(a, GPK_TUNE(SOFT_RESERVED_CUS, 32));
      if (rc) return rc;
    }
    if (kGpkExp && GPK_TUNE(LEAFK, 0)) {
      // chain-panel schedule: three bulk streams that leave LEAFK_RESERVED CUs to the chain launches (25 workgroups
      // of 147 KB LDS each).  Masked streams get hardware queues of their own, in creation order over the four
      // pipes: B -> pipe 3, Bl -> pipe 0, so one placeholder (pipe 1 = the chain's), then pipes 2, 3, 0.
      int res = GPK_TUNE(LEAFK_RESERVED, 32);
      if (res < 0 || res >= ncu) res = 0;
      rc = masked_stream(&a.Mpad, ncu, res > 0 ? res : 1, ncu);
      if (rc) return rc;
      if (GPK_TUNE(LEAFK_MB_PRIO, 0)) {  // (variant: the chain-side bulk stream unmasked at high priority)
        GPK_HIP(hipStreamCreateWithPriority(&a.Mb, hipStreamNonBlocking, hi));
      } else {
        rc = masked_stream(&a.Mb, ncu, res, ncu);
        if (rc) return rc;
      }
      rc = masked_stream(&a.Mr, ncu, res, ncu);
      if (rc) return rc;
      rc = masked_stream(&a.Mx, ncu, res, ncu);
      if (rc) return rc;
      GPK_HIP(hipMalloc((void**)&a.ck_flagL, 64));
      GPK_HIP(hipMemset(a.ck_flagL, 0, 64));
      GPK_HIP(hipMalloc((void**)&a.ck_cnt, 64));
      GPK_HIP(hipMemset(a.ck_cnt, 0, 64));
      GPK_HIP(hipExtMallocWithFlags((void**)&a.ck_flagK, 8, hipMallocSignalMemory));
      GPK_HIP(hipExtMallocWithFlags((void**)&a.ck_flagB, 8, hipMallocSignalMemory));
      GPK_HIP(hipMemset(a.ck_flagK, 0, 8));
      GPK_HIP(hipMemset(a.ck_flagB, 0, 8));
      GPK_HIP(hipDeviceSynchronize());
    }
    a.ready = true;
  }
  if (a.nev < need) {
    hipEvent_t* n = (hipEvent_t*)realloc(a.ev, sizeof(hipEvent_t) * need);
    if (!n) return GPK_E_ARG;
    a.ev = n;
    // (hipEventDisableSystemFence measured SLOWER here: 283 vs 308 steps/s on the SVGP step)
    for (int i = a.nev; i < need; ++i) GPK_HIP(hipEventCreateWithFlags(&a.ev[i], hipEventDisableTiming));
    a.nev = need;
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
                 double* invd, long strideInv, int* info, int chain_cap = 0) {
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
    g.max_wgs = chain_cap;
    rc = gpk_launch_gemm(s, g);
    if (rc) return rc;
    const int ncols = c1 - j1;
    if (ncols > 0) {
      GemmArgs u = gemm_base(below, ncols, nb, -1.0, panel, lda, panel, lda, 1.0,
                             A + (long)j1 * lda + j1, lda, batch, strideA, strideA, strideA);
      u.c_lower = 1;
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
// With `ginv` (the explicit inverse of the group's 512 x 512 diagonal block, lower, leading dimension 512; Eo != E
// required: several column tiles read what others write) the seven short dependent launches of the in-group phase --
// 64 to 192 tiles each on a 256-CU chip -- become ONE triangular-K GEMM  S = E[:,c0:c1] ginv^T.
int solve_group_fwd(hipStream_t s, const Bulk& bulk, double* E, long lde, double* Eo, long ldeo, int rows, const double* L,
                    long ldl, const double* invd, long strideInv, int n, int c0, int c1, int batch, long strideE,
                    long strideEo, long strideL, const double* ginv = nullptr) {
  int rc;
  if (ginv != nullptr) {
    GemmArgs g = gemm_base(rows, c1 - c0, c1 - c0, 1.0, E + c0, lde, ginv, c1 - c0, 0.0, Eo + c0, ldeo, batch, strideE, 0,
                           strideEo);
    g.b_tri = 2;
    bulk.apply(g);
    // (one tile per workgroup: the paired-column-tile form would put 256 tiles on 128 workgroups, half the CUs)
    if (!g.ctr) g.max_wgs = GPK_TUNE(GINV_SOLVE_WGS, 512);
    rc = gpk_launch_gemm(s, g);
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
        if (GPK_TUNE(XSMALL, 1) && batch <= 1 && !u.ctr) {
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
    bulk.apply(u);
    rc = gpk_launch_gemm(s, u);
    if (rc) return rc;
  }
  return 0;
}

// Explicit inverse of the diagonal block L[c0:c1, c0:c1] (c1 - c0 = w <= 512 columns, all panels factored) into
// ginv [w, w] (lower, leading dimension w), using wT [w, w] as scratch: the right-looking in-group solve applied to the
// rows of the identity gives L_gg^-T, which is then transposed.  Nine tiny launches (a few workgroups each) on a stream
// beside the chain; nothing here touches the bulk stream.
int group_inverse(hipStream_t s, const double* L, long ldl, const double* invd, long strideInv, int c0, int c1, double* wT,
                  double* ginv) {
  const int w = c1 - c0;
  int rc = gpk_launch_set_identity(s, wT, w, w);
  if (rc) return rc;
  // (wT - c0: the solver indexes columns globally)
  rc = solve_group_fwd(s, Bulk{}, wT - c0, w, wT - c0, w, w, L, ldl, invd, strideInv, c1, c0, c1, 1, 0, 0, 0);
  if (rc) return rc;
  return gpk_transpose((void*)s, wT, w, w, w, ginv, w, 2, 1, 0, 0);  // keep the upper triangle of L_gg^-T only
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

// ---- streamed projection (SVGP step) -----------------------------------------------------------------------------
// LTA = tril(q_sqrt)^T A needs, for output row i, the rows k >= i of A = Lm^-1 Kuf -- the LAST rows the factorisation
// produces -- so as one GEMM it can only start when everything else is over (0.55 ms of the step, alone on the chip).
// Right-looking instead: as soon as the extra-row solve has finished columns [g0,g1) of A^T, their contribution to
// ALL outputs i < g1 is added,
//     C[b, i] (+)= sum_{k in [max(i,g0), g1)} At[b,k] LqT[i,k]        (rectangle i < g0: beta = 1; triangle: beta = 0)
// on the bulk stream behind the group's solve.  The last group runs with the row-sum-of-squares epilogue (C + A B^T
// is squared, not stored).  Passed by gpk_svgp_elbo_shard to its factorisation call.
struct ProjStream {
  const double* LqT = nullptr; long ldl = 0, strideL = 0;   // [P][m, ldl]  LqT[i,k] = q_sqrt[k,i], zero for k < i
  double* C = nullptr; long ldc = 0, strideC = 0;           // [P][rows, ldc] running A^T Lq
  double* part = nullptr; long part_ld = 0, stridePart = 0; // [P][2 * tiles(m), rows] partial row sums of squares
  int P = 0;
  int groups = 0;                                           // groups issued (0 afterwards = the hook never ran)
};

int proj_group(hipStream_t s, const Bulk& bulk, ProjStream& q, const double* At, long ldat, int nrows, int g0, int g1,
               int m) {
  const bool last = g1 == m;
  const double* Ag = At + g0;
  int rc;
  if (g0 > 0) {
    GemmArgs r = gemm_base(nrows, g0, g1 - g0, 1.0, Ag, ldat, q.LqT + g0, q.ldl, 1.0, q.C, q.ldc, q.P, 0, q.strideL,
                           q.strideC);
    if (last) {
      r.epi = 1; r.sq_cols = g0; r.c2_cols = 0;
      r.part = q.part; r.part_ld = q.part_ld; r.stridePart = q.stridePart;
      r.C2 = q.part; r.ldc2 = 0; r.strideC2 = 0;
    }
    bulk.apply(r);
    rc = gpk_launch_gemm(s, r);
    if (rc) return rc;
  }
  GemmArgs t = gemm_base(nrows, g1 - g0, g1 - g0, 1.0, Ag, ldat, q.LqT + (long)g0 * q.ldl + g0, q.ldl, 0.0, q.C + g0,
                         q.ldc, q.P, 0, q.strideL, q.strideC);
  t.b_tri = 1;
  if (last) {
    t.C = nullptr;
    t.epi = 1; t.sq_cols = g1 - g0; t.c2_cols = 0;
    t.part = q.part + (long)(g0 / NB) * 2 * q.part_ld; t.part_ld = q.part_ld; t.stridePart = q.stridePart;
    t.C2 = q.part; t.ldc2 = 0; t.strideC2 = 0;
  }
  bulk.apply(t);
  rc = gpk_launch_gemm(s, t);
  if (rc) return rc;
  ++q.groups;
  return 0;
}

// cached task lists of the dataflow kernel for one shape (built and uploaded on first use: one hipMalloc + one
// synchronous hipMemcpy per new shape, at most four shapes kept)
int flow_plan(Aux& a, int n, int rows, int P, int proj, Aux::Plan** out) {
  for (int i = 0; i < a.nplans; ++i)
    if (a.plans[i].n == n && a.plans[i].rows == rows && a.plans[i].P == P && a.plans[i].proj == proj) {
      *out = &a.plans[i];
      return 0;
    }
  if (!a.flow_state) {
    GPK_HIP(hipMalloc((void**)&a.flow_state, sizeof(unsigned) * (16 + FLOW_MAX_RB)));
    GPK_HIP(hipMalloc((void**)&a.flow_flags, sizeof(unsigned) * GPK_FLOW_MAX_GROUPS));
    GPK_HIP(hipMemset(a.flow_flags, 0, sizeof(unsigned) * GPK_FLOW_MAX_GROUPS));
  }
  Aux::Plan* pl = nullptr;
  if (a.nplans < 4) {
    pl = &a.plans[a.nplans++];
  } else {  // evict the oldest (stream-ordered free is not needed: a plan is only replaced under the device mutex after
            // the work that used it was enqueued; hipFree synchronises the device)
    pl = &a.plans[0];
    if (pl->dev) (void)hipFree(pl->dev);
    pl->dev = nullptr;
  }
  pl->n = n; pl->rows = rows; pl->P = P; pl->proj = proj;
  pl->groups = flow_groups(n);
  std::vector<FlowTask> lists[8];
  flow_build(n, rows, P, proj != 0, pl->groups, lists);
  std::vector<FlowTask> all;
  pl->off[0] = 0;
  for (int x = 0; x < 8; ++x) {
    all.insert(all.end(), lists[x].begin(), lists[x].end());
    pl->off[x + 1] = (int)all.size();
  }
  GPK_HIP(hipMalloc((void**)&pl->dev, sizeof(FlowTask) * (all.size() + 1)));
  GPK_HIP(hipMemcpy(pl->dev, all.data(), sizeof(FlowTask) * all.size(), hipMemcpyHostToDevice));
  *out = pl;
  return 0;
}

// optional separate output of the solved extra rows (batch 1) + scratch for the explicit group inverses
struct ExtraOut {
  double* Eout = nullptr; long ldeout = 0;
  double* gws = nullptr;   // 2 * 512 * 512 doubles per 512-column group of the factor (gpk_ginv_ws_doubles)
};
inline size_t ginv_ws_doubles(int n) { return ((size_t)gpk_cdiv(n, NBO) + 2) * 2 * NBO * NBO; }  // one slot per group with an inverse

// ---- chain-panel schedule (A/B build; gpk_internal.h: LeafKArgs) ---------------------------------------------------
// P : one leafk launch per panel (leaf + solve of row blocks p+1, p+2 + the tiles (p+1,p+1), (p+2,p+1), (p+2,p+2)).
// Mb: [value-wait flagK(p)] solve of the rows from block p+3 on; [evR(p-1)] strip = those rows x column blocks
//     p+1..p+3 (everything panel p owes to the tiles the next chain launch touches); [write flagB(p)].
// Mr: [evS(p)] rest-update: columns from block p+4 on, lower tiles.
// X : the extra rows, group by group, behind evS of the group's last panel (as in the stream schedule).
// No packet is ever put on P between two chain launches.
int potrf_leafk(Aux* aux, hipStream_t S, double* A, int n, int extra, long lda, double* invd, int zero_upper, int* info,
                const ExtraOut* xo, const std::function<int(hipStream_t)>* x_prologue, int tri) {
  const int nblk = n / NB;
  const long strideInv = (long)nblk * NB * NB;
  const bool ride = extra > 0 && extra <= 256;
  const int R = ride ? n + extra : n;
  const bool useX = extra > 0 && !ride;
  double* E = A + (long)n * lda;
  const bool oop = xo && xo->Eout && useX;
  double* Eo = oop ? xo->Eout : E;
  const long ldeo = oop ? xo->ldeout : lda;
  const bool use_ginv = oop && xo->gws && GPK_TUNE(GROUP_INVERSE, 0);
  hipStream_t P = aux->P, Mb = aux->Mb, Mr = aux->Mr, X = aux->Mx;
  hipEvent_t* evS = aux->ev;
  hipEvent_t* evR = aux->ev + nblk;
  hipEvent_t* evG = aux->ev + 2 * nblk;
  hipEvent_t evFork = aux->ev[3 * nblk], evJoinP = aux->ev[3 * nblk + 1], evJoinB = aux->ev[3 * nblk + 2],
             evJoinX = aux->ev[3 * nblk + 3], evJoinR = aux->ev[3 * nblk + 4];
  int rc;
  if (x_prologue && !useX) {
    rc = (*x_prologue)(S);
    if (rc) return rc;
  }
  if (aux->ck_epoch > 0x7fff0000u) {  // (epochs are compared with >=: start over long before they wrap)
    GPK_HIP(hipDeviceSynchronize());
    GPK_HIP(hipMemset(aux->ck_flagL, 0, 64));
    GPK_HIP(hipMemset(aux->ck_flagK, 0, 8));
    GPK_HIP(hipMemset(aux->ck_flagB, 0, 8));
    GPK_HIP(hipDeviceSynchronize());
    aux->ck_epoch = 0;
  }
  const unsigned base = aux->ck_epoch;
  aux->ck_epoch += (unsigned)nblk + 1u;
  GPK_HIP(hipEventRecord(evFork, S));
  GPK_HIP(hipStreamWaitEvent(P, evFork, 0));
  GPK_HIP(hipStreamWaitEvent(Mb, evFork, 0));
  GPK_HIP(hipStreamWaitEvent(Mr, evFork, 0));
  if (useX) GPK_HIP(hipStreamWaitEvent(X, evFork, 0));
  Bulk bulk;  // (uncapped: the bulk streams cannot reach the chain's CUs)
  bulk.cap = GPK_TUNE(LEAFK_EXTRA_MAX_WGS, 0);
  int last_rest = -1, xg0 = 0;
  const int xgroup = std::max(NB, (GPK_TUNE(XGROUP, NBO) / NB) * NB);
  for (int p = 0; p < nblk; ++p) {
    const int c0 = p * NB, c1 = c0 + NB;
    const bool b1 = p + 1 < nblk, b2 = p + 2 < nblk;
    LeafKArgs a{};
    a.A = A; a.lda = lda; a.nblk = nblk; a.p = p; a.invd = invd; a.info = info;
    a.flagL = aux->ck_flagL; a.cnt = aux->ck_cnt; a.flagK = aux->ck_flagK; a.flagB = aux->ck_flagB;
    a.epoch = base + (unsigned)p + 1u;
    a.need_b = p > 0 ? base + (unsigned)p : 0u;
    a.nh = b1 ? (b2 ? 24 : 8) : 0;
    a.nsolve = b1 ? (b2 ? 16 : 8) : 0;
    rc = gpk_launch_leafk(P, a);
    if (rc) return rc;
    // ---- Mb: the other rows of panel p, then what panel p owes to the next chain launch's tiles -----------------
    GPK_HIP(hipStreamWaitValue32(Mb, aux->ck_flagK, a.epoch, hipStreamWaitValueGte, 0xffffffffu));
    const int rs = std::min((p + 3) * NB, n);  // first row of the bulk side (blocks p+1, p+2 belong to the chain launch)
    const double* invp = invd + (long)p * NB * NB;
    if (R > rs) {
      double* panel = A + (long)rs * lda + c0;
      GemmArgs g = gemm_base(R - rs, NB, NB, 1.0, panel, lda, invp, NB, 0.0, panel, lda, 1, 0, 0, 0);
      g.b_tri = 2;
      rc = gpk_launch_gemm(Mb, g);
      if (rc) return rc;
    }
    GPK_HIP(hipEventRecord(evS[p], Mb));
    const int cE = std::min((p + 4) * NB, n);
    if (R > rs && cE > c1) {
      if (last_rest >= 0) GPK_HIP(hipStreamWaitEvent(Mb, evR[last_rest], 0));  // (column block p+3 also got rest-update p-1)
      GemmArgs u = gemm_base(R - rs, cE - c1, NB, -1.0, A + (long)rs * lda + c0, lda, A + (long)c1 * lda + c0, lda, 1.0,
                             A + (long)rs * lda + c1, lda, 1, 0, 0, 0);
      rc = gpk_launch_gemm(Mb, u);
      if (rc) return rc;
    }
    GPK_HIP(hipStreamWriteValue32(Mb, aux->ck_flagB, a.epoch, 0));
    // ---- Mr: rest of the trailing update, columns from block p+4 on ------------------------------------------------
    const int c4 = (p + 4) * NB;
    if (c4 < n) {
      GPK_HIP(hipStreamWaitEvent(Mr, evS[p], 0));
      const double* P4 = A + (long)c4 * lda + c0;
      GemmArgs u = gemm_base(R - c4, n - c4, NB, -1.0, P4, lda, P4, lda, 1.0, A + (long)c4 * lda + c4, lda, 1, 0, 0, 0);
      u.c_lower = 1;
      rc = gpk_launch_gemm(Mr, u);
      if (rc) return rc;
      GPK_HIP(hipEventRecord(evR[p], Mr));
      last_rest = p;
    }
    if (p == 0 && x_prologue && useX) {
      rc = (*x_prologue)(X);
      if (rc) return rc;
    }
    // ---- X: the extra rows against the finished columns (same grouping as the stream schedule) -----------------------
    const bool tail_zone = n >= 8 * NB;
    const bool tail_group = tail_zone && (c1 == n - 2 * NB || c1 == n - NB);
    const bool full_group = (c1 - xg0) >= xgroup && !(tail_zone && c1 > n - 2 * NB && c1 < n);
    if (useX && (c1 == n || full_group || tail_group)) {
      const int g0 = xg0;
      xg0 = c1;
      const double* ginv = nullptr;
      if (use_ginv && c1 - g0 == NBO && (g0 % NBO) == 0) {
        double* wT = xo->gws + (size_t)(g0 / NBO) * 2 * NBO * NBO;
        double* gi = wT + (size_t)NBO * NBO;
        GPK_HIP(hipStreamWaitEvent(Mr, evS[p], 0));
        rc = group_inverse(Mr, A, lda, invd, strideInv, g0, c1, wT, gi);
        if (rc) return rc;
        GPK_HIP(hipEventRecord(evG[p], Mr));
        GPK_HIP(hipStreamWaitEvent(X, evG[p], 0));
        ginv = gi;
      }
      GPK_HIP(hipStreamWaitEvent(X, evS[p], 0));
      for (int h0 = g0; h0 < c1; h0 += NBO) {
        const int h1 = std::min(h0 + NBO, c1);
        const int xrows = tri ? extra - tri + h1 : extra;
        rc = solve_group_fwd(X, bulk, E, lda, Eo, ldeo, xrows, A, lda, invd, strideInv, n, h0, h1, 1, 0, 0, 0, ginv);
        if (rc) return rc;
      }
    }
  }
  GPK_HIP(hipEventRecord(evJoinP, P));
  GPK_HIP(hipStreamWaitEvent(S, evJoinP, 0));
  GPK_HIP(hipEventRecord(evJoinB, Mb));
  GPK_HIP(hipStreamWaitEvent(S, evJoinB, 0));
  GPK_HIP(hipEventRecord(evJoinR, Mr));
  GPK_HIP(hipStreamWaitEvent(S, evJoinR, 0));
  if (useX) {
    GPK_HIP(hipEventRecord(evJoinX, X));
    GPK_HIP(hipStreamWaitEvent(S, evJoinX, 0));
  }
  if (zero_upper) return gpk_launch_zero_upper(S, A, n, lda, 1, 0);
  return 0;
}

// x_prologue: work of the CALLER that belongs on the bulk stream before the first extra-row group (the SVGP driver's Kfu
// build, transposes, KL).  It is enqueued after the first panel's chain kernels: every host call issued before the first
// leaf delays the whole step, and nothing on the bulk stream is needed for ~4 panels.
int potrf_core(hipStream_t S, double* A, int n, int extra, long lda, int batch, long strideA, double* invd, int zero_upper,
               int* info, ProjStream* proj, const ExtraOut* xo = nullptr,
               const std::function<int(hipStream_t)>* x_prologue = nullptr, int tri = 0) {
  if (!A || !invd || n < 0 || extra < 0 || lda < n) return GPK_E_ARG;
  if (tri && (tri != n || extra < n || batch > 1)) return GPK_E_ARG;
  if (batch <= 0) batch = 1;
  if (info) GPK_HIP(hipMemsetAsync(info, 0, sizeof(int) * batch, S));
  if (n == 0) return 0;
  // tri = n: the LAST n extra rows are the identity (written here) and come back as L^-T.  Row j of that block stays
  // zero left of column j, so column group [c0, c1) only has to process its first c1 rows: n^3 / 3 flop instead of n^3.
  if (tri) {
    const int rci = gpk_launch_set_identity(S, A + (long)(n + extra - tri) * lda, n, lda);
    if (rci) return rci;
  }
  const long strideInv = (long)gpk_cdiv(n, NB) * NB * NB;
  // outer panel width for n >= 4096: A/B at N = 16384 (same box): 384 -> 34.6 ms, 512 -> 33.2, 640 -> 32.6, 768 -> 32.4,
  // 896 -> 32.3, 1024 -> 32.7; one leaf block for the SVGP sizes, where the whole factorisation is a latency chain
  const int nbo_large = (GPK_TUNE(NBO, 640) / NB) * NB;
  const int nbo = (n >= 4096) ? (nbo_large >= NB ? nbo_large : NBO) : NB;
  const int npanels = gpk_cdiv(n, nbo);
  // Few extra rows (GPR: the P columns of Y) simply ride along through the panel solves and trailing
  // updates of the square part; many extra rows (SVGP: the minibatch; GPR: the test rows of predict_f) are solved
  // right-looking, group by group, as bulk work overlapped with the factorisation.
  const bool ride = extra > 0 && extra <= 256;
  const int R = ride ? n + extra : n;  // rows handled together with the square part
  const bool useX = extra > 0 && !ride;
  double* E = A + (long)n * lda;       // the extra rows
  // solved extra rows: in place, or in the caller's separate matrix (only when they are solved apart from the square part)
  const bool oop = xo && xo->Eout && useX && batch == 1;
  double* Eo = oop ? xo->Eout : E;
  const long ldeo = oop ? xo->ldeout : lda;
  const long strideEo = oop ? 0 : strideA;
  int rc;
  if (n <= NB) {  // one leaf; nothing to overlap
    if (x_prologue) {
      rc = (*x_prologue)(S);
      if (rc) return rc;
    }
    rc = factor_panel(S, A, R, 0, n, lda, batch, strideA, invd, strideInv, info);
    if (rc) return rc;
    if (useX) {
      rc = solve_group_fwd(S, Bulk{}, E, lda, Eo, ldeo, extra, A, lda, invd, strideInv, n, 0, n, batch, strideA, strideEo,
                           strideA);
      if (rc) return rc;
    }
    return zero_upper ? gpk_launch_zero_upper(S, A, n, lda, batch, strideA) : 0;
  }
  int dev = 0;
  rc = current_device(&dev);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lock(g_aux[dev].mu);
  Aux* aux = nullptr;
  rc = aux_get(dev, 3 * npanels + 8, &aux);
  if (rc) return rc;
  const bool large = n >= 4096;
  if (kGpkExp && GPK_TUNE(LEAFK, 0) && aux->Mb && !large && batch == 1 && (n % NB) == 0 && n >= 3 * NB && !proj) {
    return potrf_leafk(aux, S, A, n, extra, lda, invd, zero_upper, info, xo, x_prologue, tri);
  }
  hipStream_t P = aux->P, B = large ? aux->B : aux->Bs;
  // ONE bulk stream beside the chain: for large factorisations the extra rows share the (hardware-masked) stream of the
  // trailing updates; for small ones they have stream X, whose GEMMs are ticketed under the software reservation.
  hipStream_t X = large ? aux->B : aux->X;
  Bulk bulk;
  if (kGpkExp && !large && batch == 1 && GPK_TUNE(SOFT_RESERVE, 0)) {
    bulk.ctr = aux->ctr;
    bulk.resv = aux->resv_cus > 0 ? aux->resv : nullptr;
    bulk.min_k = GPK_TUNE(RESERVE_MIN_K, 256);
  } else if (!large) {
    // cap on the persistent workgroups of the big extra-row updates, so that some CUs stay free for the panel stream's
    // leaf kernel (A/B on the SVGP step: cap 320 -> 448 steps/s, no cap 435, cap 224 -> 431)
    bulk.cap = GPK_TUNE(EXTRA_MAX_WGS, 320);
  }
  hipEvent_t* evF = aux->ev;            // [npanels] panel p factored, rows below solved (recorded on P)
  hipEvent_t* evR = aux->ev + npanels;  // [npanels] rest of the trailing update of panel p done (on B)
  hipEvent_t* evG = aux->ev + 2 * npanels;  // [npanels] explicit inverse of the group ending with panel p ready (on B)
  hipEvent_t evFork = aux->ev[3 * npanels], evJoinP = aux->ev[3 * npanels + 1], evJoinB = aux->ev[3 * npanels + 2],
             evJoinX = aux->ev[3 * npanels + 3], evLate = aux->ev[3 * npanels + 4];
  const bool use_ginv = kGpkExp && oop && xo->gws && !large && GPK_TUNE(GROUP_INVERSE, 0);
  // ---- SVGP sizes with a separate output matrix: the whole bulk side is ONE dataflow launch on X ----------------------
  const bool use_flow = use_ginv && (n % NB) == 0 && n >= 4 * NB && n / NB <= 255 && gpk_cdiv(extra, NB) <= FLOW_MAX_RB &&
                        (!proj || proj->P <= 255) && GPK_TUNE(FLOW, 0);
  Aux::Plan* plan = nullptr;
  // (the streamed projection only pays inside the dataflow launch; as separate K <= 512 read-modify-write launches on
  //  the bulk stream it measured 3.46 vs 2.59 ms per step, so without a plan the caller projects afterwards)
  if (proj && !use_flow && !GPK_TUNE(STREAM_PROJ_MULTI, 0)) proj = nullptr;
  if (use_flow) {
    rc = flow_plan(*aux, n, extra, proj ? proj->P : 0, proj ? 1 : 0, &plan);
    if (rc) return rc;
    if ((int)plan->groups.size() > GPK_FLOW_MAX_GROUPS) plan = nullptr;
  }
  unsigned epoch = 0;
  FlowArgs flow{};
  if (plan) {
    epoch = ++aux->epoch;
    if (epoch == 0) epoch = ++aux->epoch;
    FlowArgs f{};
    f.E = E; f.lde = lda; f.Eo = Eo; f.ldeo = ldeo; f.L = A; f.ldl = lda; f.invd = invd; f.gws = xo->gws;
    if (proj) {
      f.LqT = proj->LqT; f.ldq = proj->ldl; f.strideQ = proj->strideL;
      f.Cacc = proj->C; f.ldc = proj->ldc; f.strideC = proj->strideC;
      f.part = proj->part; f.part_ld = proj->part_ld; f.stridePart = proj->stridePart;
      f.P = proj->P;
      proj->groups = (int)plan->groups.size();
    }
    f.rows = extra; f.n = n; f.ng = (int)plan->groups.size();
    for (int i = 0; i < f.ng; ++i) { f.g0[i] = plan->groups[i].g0; f.g1[i] = plan->groups[i].g1; f.ginv[i] = plan->groups[i].ginv; }
    f.tasks = plan->dev;
    for (int i = 0; i < 9; ++i) f.off[i] = plan->off[i];
    f.ctr = aux->flow_state; f.prog = aux->flow_state + 16; f.flags = aux->flow_flags; f.epoch = epoch;
    f.resv = (GPK_TUNE(SOFT_RESERVE, 0) && aux->resv_cus > 0) ? aux->resv : nullptr;
    f.info = info;
    f.coh = GPK_TUNE(FLOW_COH, 2);  // 2: sc1 stores + sc1 accumulator preload + L1 invalidate before solve tasks; 5: ordinary accesses + one L1 invalidate per task
    flow = f;
  }
  if (x_prologue && !useX) {  // the extra rows ride through the panel solves: they must exist before the first one
    rc = (*x_prologue)(S);
    if (rc) return rc;
  }
  GPK_HIP(hipEventRecord(evFork, S));  // fork: everything already queued on S comes first
  GPK_HIP(hipStreamWaitEvent(P, evFork, 0));
  if (B != S) GPK_HIP(hipStreamWaitEvent(B, evFork, 0));
  if (useX && X != B) GPK_HIP(hipStreamWaitEvent(X, evFork, 0));
  // The persistent bulk kernel is launched when the FIRST column group's flag has been raised: until then the chain
  // has the whole chip for its largest panels, afterwards it is confined to the reserved compute units.
  bool flow_started = false;
  const int chain_wgs = std::max(8, GPK_TUNE(CHAIN_WGS, 0) > 0 ? GPK_TUNE(CHAIN_WGS, 0) : aux->resv_cus);
  if (plan) GPK_HIP(hipMemsetAsync(aux->flow_state, 0, sizeof(unsigned) * (16 + gpk_cdiv(extra, NB)), X));
  hipStream_t last_bulk = B;
  int last_rest = -1;  // panel index whose evR marks the most recent rest-update
  int xg0 = 0;         // first column of the current extra-row group
  const int late_rows = GPK_TUNE(LATE_ROWS, 3072);  // A/B at N = 16384: 3072 -> 32.9 ms, 6144 -> 33.8, off -> 33.1
  const int xgroup = std::max(NB, (GPK_TUNE(XGROUP, NBO) / NB) * NB);
  for (int p = 0; p < npanels; ++p) {
    const int c0 = p * nbo;
    const int c1 = (c0 + nbo < n) ? c0 + nbo : n;
    const int c2 = (c1 + nbo < n) ? c1 + nbo : n;
    // ---- P: the critical path.  Panel p, then the strip = columns of panel p+1 (look-ahead) -----------
    // (dataflow mode: once the bulk kernel is resident the chain only finds the reserved compute units free, so its
    //  one-shot GEMMs are launched as ONE round of workgroups that walk the row blocks)
    const int chain_cap = (plan && flow_started) ? chain_wgs : 0;
    rc = factor_panel(P, A, R, c0, c1, lda, batch, strideA, invd, strideInv, info, chain_cap);
    if (rc) return rc;
    GPK_HIP(hipEventRecord(evF[p], P));
    const double* Pn = A + (long)c1 * lda + c0;  // rows c1.. of the solved panel
    if (c1 < n) {
      // columns c1:c2 also received the most recent rest-update (on a bulk stream): order the two
      if (last_rest >= 0) GPK_HIP(hipStreamWaitEvent(P, evR[last_rest], 0));
      GemmArgs u = gemm_base(R - c1, c2 - c1, c1 - c0, -1.0, Pn, lda, Pn, lda, 1.0,
                             A + (long)c1 * lda + c1, lda, batch, strideA, strideA, strideA);
      u.c_lower = 1;
      u.max_wgs = chain_cap;
      rc = gpk_launch_gemm(P, u);
      if (rc) return rc;
    }
    // ---- B: rest of the outer trailing update  A[c2:, c2:] -= P[c2:] P[c2:]^T, lower tiles only --------
    // While the trailing matrix is large the factorisation is bound by these GEMMs and they start as soon as
    // panel p is solved.  Near the end it is bound by the latency chain of P instead: there the strip goes
    // first (alone on the chip) and the rest-update overlaps the NEXT panel's chain rather than the strip.
    const bool strip_first = large && (n - c1 <= late_rows) && (c1 < n);
    if (c2 < n) {
      hipStream_t Bp = B;
      if (strip_first) {
        Bp = aux->Bl;  // (in-order with the earlier rest-updates through evR below)
        GPK_HIP(hipEventRecord(evLate, P));
        GPK_HIP(hipStreamWaitEvent(Bp, evLate, 0));
        if (last_rest >= 0) GPK_HIP(hipStreamWaitEvent(Bp, evR[last_rest], 0));
      } else if (plan && flow_started && GPK_TUNE(REST_AFTER_STRIP, 1)) {
        // the chain is confined to the reserved CUs: a rest-update that starts together with the strip fights it for
        // them (strip 15 -> 50 us, rest-update 25 -> 85 us in the timeline); behind the strip it overlaps the next leaf
        GPK_HIP(hipEventRecord(evLate, P));
        GPK_HIP(hipStreamWaitEvent(B, evLate, 0));
      } else {
        GPK_HIP(hipStreamWaitEvent(B, evF[p], 0));
      }
      const double* P2 = A + (long)c2 * lda + c0;
      GemmArgs u = gemm_base(R - c2, n - c2, c1 - c0, -1.0, P2, lda, P2, lda, 1.0,
                             A + (long)c2 * lda + c2, lda, batch, strideA, strideA, strideA);
      u.c_lower = 1;
      if (Bp == aux->B && large) u.stagger_first = aux->bulk_cus;
      if (plan && flow_started) u.no_small = 1;  // (150 KB one-shot workgroups: one per reserved CU -> many rounds)
      rc = gpk_launch_gemm(Bp, u);
      if (rc) return rc;
      GPK_HIP(hipEventRecord(evR[p], Bp));
      last_bulk = Bp;
      last_rest = p;
    }
    if (p == 0 && x_prologue && useX) {
      rc = (*x_prologue)(X);
      if (rc) return rc;
    }
    // ---- X: the extra rows against the finished columns, in groups of up to 512 columns (so that the big
    // right-looking update is a K = 512 GEMM).  For the small sizes the groups shrink towards the end (.., n-256,
    // n-128, n): whatever is left of the extra-row work when the LAST leaf finishes is exposed latency.
    const bool tail_zone = (nbo == NB) && (n >= 8 * NB);
    const bool tail_group = tail_zone && (c1 == n - 2 * NB || c1 == n - NB);
    const bool full_group = ((c1 - xg0) >= xgroup || (large && c1 - xg0 >= nbo)) && !(tail_zone && c1 > n - 2 * NB && c1 < n);
    if (plan) {
      // dataflow mode: all this stream code has to do is raise the flag of the column group that ends with this panel,
      // on B behind the rest-update (and, for a 512-column group, behind the assembly of its explicit inverse)
      for (size_t gi = 0; gi < plan->groups.size(); ++gi) {
        const FlowGroup& fg = plan->groups[gi];
        if (fg.g1 != c1) continue;
        if (c2 >= n) GPK_HIP(hipStreamWaitEvent(B, evF[p], 0));  // (no rest-update was issued for this panel)
        if (fg.ginv) {
          double* wT = xo->gws + gi * 2 * NBO * NBO;
          rc = group_inverse(B, A, lda, invd, strideInv, fg.g0, fg.g1, wT, wT + (size_t)NBO * NBO);
          if (rc) return rc;
        }
        rc = gpk_launch_set_flag(B, aux->flow_flags + gi, epoch);
        if (rc) return rc;
        if (!flow_started) {
          GPK_HIP(hipEventRecord(evG[p], B));
          GPK_HIP(hipStreamWaitEvent(X, evG[p], 0));
          rc = gpk_launch_flow(X, flow);
          if (rc) return rc;
          flow_started = true;
        }
      }
    } else if (useX && (c1 == n || full_group || tail_group)) {
      const int g0 = xg0;
      xg0 = c1;
      // a full, aligned 512-column group: its explicit inverse is assembled on B (right behind the rest-update of the
      // group's last panel; a handful of one-to-four-workgroup launches) and the bulk stream solves the group in one GEMM
      const double* ginv = nullptr;
      if (use_ginv && c1 - g0 == NBO && (g0 % NBO) == 0) {
        double* wT = xo->gws + (size_t)(g0 / NBO) * 2 * NBO * NBO;
        double* gi = wT + (size_t)NBO * NBO;
        if (c2 >= n) GPK_HIP(hipStreamWaitEvent(B, evF[p], 0));  // (no rest-update was issued for this panel)
        rc = group_inverse(B, A, lda, invd, strideInv, g0, c1, wT, gi);
        if (rc) return rc;
        GPK_HIP(hipEventRecord(evG[p], B));
        GPK_HIP(hipStreamWaitEvent(X, evG[p], 0));
        ginv = gi;
      }
      GPK_HIP(hipStreamWaitEvent(X, evF[p], 0));
      // (columns [g0, c1) may span several 512-groups when the outer panel is wider than a group)
      for (int h0 = g0; h0 < c1; h0 += NBO) {
        const int h1 = std::min(h0 + NBO, c1);
        const int xrows = tri ? extra - tri + h1 : extra;  // (identity rows below column h1 are still exactly zero here)
        rc = solve_group_fwd(X, bulk, E, lda, Eo, ldeo, xrows, A, lda, invd, strideInv, n, h0, h1, batch, strideA,
                             strideEo, strideA, ginv);
        if (rc) return rc;
        if (proj) {
          rc = proj_group(X, bulk, *proj, Eo, ldeo, extra, h0, h1, n);
          if (rc) return rc;
        }
      }
    }
  }
  // join: P has waited for every rest-update it depends on; B's last event covers the rest
  GPK_HIP(hipEventRecord(evJoinP, P));
  GPK_HIP(hipStreamWaitEvent(S, evJoinP, 0));
  if (plan) last_bulk = B;  // (flag kernels follow the last rest-update on B)
  if (last_bulk != S) {
    GPK_HIP(hipEventRecord(evJoinB, last_bulk));  // rest-updates are chained through evR, the last one covers all
    GPK_HIP(hipStreamWaitEvent(S, evJoinB, 0));
  }
  if (useX && X != last_bulk) {
    GPK_HIP(hipEventRecord(evJoinX, X));
    GPK_HIP(hipStreamWaitEvent(S, evJoinX, 0));
  }
  if (zero_upper) return gpk_launch_zero_upper(S, A, n, lda, batch, strideA);
  return 0;
}
}  // namespace

extern "C" int gpk_potrf(void* stream, double* A, int n, int extra, long lda, int batch,
                         long strideA, double* invd, int zero_upper, int* info) {
  return potrf_core((hipStream_t)stream, A, n, extra, lda, batch, strideA, invd, zero_upper, info, nullptr);
}

extern "C" int gpk_potrf_inv(void* stream, double* A, int n, int extra, long lda, double* invd, int zero_upper,
                             int* info) {
  return potrf_core((hipStream_t)stream, A, n, extra + n, lda, 1, 0, invd, zero_upper, info, nullptr, nullptr, nullptr, n);
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
    for (int g0 = 0; g0 < n; g0 += NBO) {
      rc = solve_group_fwd(s, Bulk{}, B, ldb, B, ldb, m, L, ldl, invd, strideInv, n, g0, std::min(g0 + NBO, n), batch,
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

extern "C" int gpk_project(void* stream, const double* At, int rows, int m, long ldat,
                           const double* LqT, long ldl, int P, double* ssq, void* ws,
                           size_t ws_bytes) {
  if (!At || !LqT || !ssq || rows < 0 || m <= 0 || P <= 0) return GPK_E_ARG;
  if (!ws || ws_bytes < gpk_project_workspace_bytes(rows, m, P)) return GPK_E_WORKSPACE;
  if (rows == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int nt = 2 * gpk_gemm_tiles_n(m);
  GemmArgs g = gemm_base(rows, m, m, 1.0, At, ldat, LqT, ldl, 0.0, nullptr, 0, P, 0, (long)m * ldl, 0);
  g.b_tri = 1;  // LqT[j,k] = Lq[k,j] vanishes for k < j
  g.epi = 1; g.sq_cols = m; g.c2_cols = 0;
  g.part = (double*)ws; g.part_ld = rows; g.stridePart = (long)nt * rows;
  g.C2 = (double*)ws; g.ldc2 = 0; g.strideC2 = 0;
  int rc = gpk_launch_gemm(s, g);
  if (rc) return rc;
  return gpk_launch_sum_parts(s, (const double*)ws, nt, rows, (long)nt * rows, P, ssq);
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
                           double variance, double noise_variance, double mean_const, double* out,
                           int* info, void* ws, size_t ws_bytes) {
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
  // K(X,X) + noise I, lower tiles only (gpr.py:100-101)
  rc = gpk_kernel_matrix(stream, family, X, n, ldx, nullptr, 0, 0, d, ls_host, ard, variance,
                         noise_variance, 1, T, l.ld);
  if (rc) return rc;
  // (Y - m)^T as P extra rows (gpr.py:103, logdensities.py:149)
  rc = gpk_launch_transpose_shift(s, Y, n, P, ldy, T + (long)n * l.ld, l.ld, -mean_const);
  if (rc) return rc;
  // L = chol(K); extra rows -> alpha^T = (L^-1 (Y-m))^T  (gpr.py:102, logdensities.py:150)
  rc = gpk_potrf(stream, T, n, P, l.ld, 1, 0, invd, 0, info);
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
#ifndef GPK_STREAM_PROJ_DEFAULT
#define GPK_STREAM_PROJ_DEFAULT 0   // (A/B build: 1 makes the projection part of the dataflow launch; ignored without it)
#endif
namespace {
struct ElboLayout {
  long ld; int nt;
  size_t off_T, off_invd, off_LqT, off_s0, off_fmean, off_ssq, off_proj, off_part0, off_part1, off_C, off_At, off_gws, total;
};
// the q_sqrt projection streamed behind the extra-row solve (1) or as one GEMM after the factorisation (0)
inline bool stream_proj_on() { return kGpkExp && GPK_TUNE(STREAM_PROJ, GPK_STREAM_PROJ_DEFAULT) != 0; }
// a separate A^T matrix + group-inverse scratch are only needed by the A/B-build schemes that solve whole column groups
inline bool separate_at_on() { return kGpkExp && (GPK_TUNE(GROUP_INVERSE, 0) || GPK_TUNE(FLOW, 0)); }

ElboLayout elbo_layout(int m, int rows, int P, int q_diag) {
  ElboLayout l{};
  l.ld = (long)gpk_align_up((size_t)m, 8);
  l.nt = 2 * gpk_gemm_tiles_n(m);
  size_t o = 0;
  l.off_T = o; o += gpk_align_up((size_t)(m + rows) * l.ld * sizeof(double), 256);
  l.off_invd = o; o += gpk_align_up(gpk_invd_elems(m, 1) * sizeof(double), 256);
  l.off_LqT = o; o += q_diag ? 0 : gpk_align_up((size_t)P * m * l.ld * sizeof(double), 256);
  l.off_s0 = o; o += gpk_align_up((size_t)rows * sizeof(double), 256);
  l.off_fmean = o; o += gpk_align_up((size_t)rows * P * sizeof(double), 256);
  l.off_ssq = o; o += gpk_align_up((size_t)rows * P * sizeof(double), 256);
  l.off_proj = o; o += q_diag ? 0 : gpk_align_up(gpk_project_workspace_bytes(rows, m, P), 256);
  l.off_part0 = o; o += gpk_align_up((size_t)GPK_REDUCE_MAXPART * sizeof(double), 256);
  l.off_part1 = o; o += gpk_align_up((size_t)GPK_REDUCE_MAXPART * sizeof(double), 256);
  l.off_C = o; o += (q_diag || !stream_proj_on()) ? 0 : gpk_align_up((size_t)P * rows * l.ld * sizeof(double), 256);  // running A^T Lq (streamed projection)
  l.off_At = o; o += separate_at_on() ? gpk_align_up((size_t)rows * l.ld * sizeof(double), 256) : 0;  // A^T apart from the consumed Kfu rows
  l.off_gws = o; o += separate_at_on() ? gpk_align_up(ginv_ws_doubles(m) * sizeof(double), 256) : 0;  // explicit group inverses
  l.total = o;
  return l;
}
}  // namespace

extern "C" size_t gpk_svgp_elbo_workspace_bytes(int m, int rows, int d, int P, int q_diag) {
  (void)d;
  return elbo_layout(m, rows, P, q_diag).total;
}

extern "C" int gpk_svgp_elbo_shard(void* stream, int family, const double* Z, int m, long ldz,
                                   const double* Xb, const double* Yb, int rows, long ldxb,
                                   long ldyb, int d, int P, const double* ls_host, int ard,
                                   double variance, double noise_variance, double jitter,
                                   double mean_const, const double* q_mu, const double* q_sqrt,
                                   int q_diag, int whiten, double* out, int* info, void* ws,
                                   size_t ws_bytes) {
  if (!Z || !Xb || !Yb || !q_mu || !q_sqrt || !out || !info || m <= 0 || rows < 0 || P <= 0 || P > 16)
    return GPK_E_ARG;
  if (!whiten) return GPK_E_UNSUPPORTED;  // composed from the primitives by the Python host
  const ElboLayout l = elbo_layout(m, rows, P, q_diag);
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
  double* Kfu = T + (long)m * l.ld;  // extra rows of the trapezoid: Kfu, consumed by the factorisation
  int rc;
  // Kuf^T = k(Xb, Z) as the extra rows (posteriors.py:836, covariances/kufs.py:31-34).  Only the bulk stream of the
  // factorisation consumes it, so it is built THERE (ordered after everything already queued on the caller's stream)
  // and the panel chain starts right after the much smaller Kuu build.  Work that depends on neither factorisation
  // nor minibatch solve -- tril(q_sqrt)^T for the projection and the whole KL term -- goes to that stream too, which
  // idles until the first 512 columns of Lm exist; gpk_potrf joins it.
  const bool side = m > GPK_NB && m < 4096 && rows > 256;
  // A^T = Kfu Lm^-T: in its own matrix when the extra rows are solved apart from the square part (then whole 512-column
  // groups are solved with one GEMM against the group's explicit inverse), in place otherwise
  const bool sep = side && separate_at_on();
  double* At = sep ? (double*)(w + l.off_At) : Kfu;
  int dev = 0;
  rc = current_device(&dev);
  if (rc) return rc;
  // Kuu + jitter I (posteriors.py:835, covariances/kuus.py:29-34), lower tiles only: the chain's first leaf waits for
  // nothing else, so it is the first thing enqueued
  rc = gpk_kernel_matrix(stream, family, Z, m, ldz, nullptr, 0, 0, d, ls_host, ard, variance, jitter,
                         1, T, l.ld);
  if (rc) return rc;
  int c1 = 0;
  // everything else that precedes the minibatch solve, as one closure: enqueued by the factorisation on its bulk stream
  // (side) or here on the caller's stream
  const std::function<int(hipStream_t)> prologue = [&](hipStream_t xs) -> int {
    int r = gpk_kernel_matrix((void*)xs, family, Xb, rows, ldxb, Z, m, ldz, d, ls_host, ard, variance, 0.0, 0, Kfu, l.ld);
    if (r) return r;
    if (!side) return 0;
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
  // Lm = chol(Kuu);  A^T = Kfu Lm^-T   (conditionals/util.py:67,125); optionally the projection rides along
  ProjStream ps;
  const bool stream_proj = stream_proj_on() && side && !q_diag && (m % GPK_NB) == 0;
  if (stream_proj) {
    ps.LqT = LqT; ps.ldl = l.ld; ps.strideL = (long)m * l.ld;
    ps.C = (double*)(w + l.off_C); ps.ldc = l.ld; ps.strideC = (long)rows * l.ld;
    ps.part = (double*)(w + l.off_proj); ps.part_ld = rows; ps.stridePart = (long)l.nt * rows;
    ps.P = P;
  }
  ExtraOut xo;
  if (sep) {
    xo.Eout = At; xo.ldeout = l.ld; xo.gws = (double*)(w + l.off_gws);
  }
  rc = potrf_core(s, T, m, rows, l.ld, 1, 0, invd, 0, info, stream_proj ? &ps : nullptr, sep ? &xo : nullptr, &prologue);
  if (rc) return rc;
  const bool projected = stream_proj && ps.groups > 0;
  // s0 = sum_k A^2 (util.py:133), fmean = A^T q_mu (util.py:144), q_diag: ssq = sum (A q_sqrt)^2 (:149)
  rc = gpk_row_stats(stream, At, rows, m, l.ld, q_mu, q_diag ? q_sqrt : nullptr, P, 1.0, 0.0, s0, fmean,
                     q_diag ? ssq : nullptr);
  if (rc) return rc;
  if (!q_diag) {
    // L = band_part(q_sqrt,-1,0); LTA = L^T A; ssq = sum LTA^2   (util.py:151-164)
    if (!side) {
      rc = gpk_transpose(stream, q_sqrt, m, m, m, LqT, l.ld, 1, P, (long)m * m, (long)m * l.ld);
      if (rc) return rc;
    }
    if (projected) {
      rc = gpk_launch_sum_parts(s, (const double*)(w + l.off_proj), l.nt, rows, (long)l.nt * rows, P, ssq);
    } else {
      rc = gpk_project(stream, At, rows, m, l.ld, LqT, l.ld, P, ssq, w + l.off_proj,
                       gpk_project_workspace_bytes(rows, m, P));
    }
    if (rc) return rc;
  }
  // sum_b var_exp_b  (likelihoods/scalar_continuous.py:139-148, svgp.py:174,181)
  int c0 = 0;
  rc = gpk_launch_varexp_stage1(s, Yb, ldyb, fmean, rows, P, s0, 0, ssq, &variance, 0, noise_variance,
                                mean_const, nullptr, part0, &c0);
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
