// Host-side orchestration (no device code here): the trapezoidal blocked Cholesky, triangular solves
// against a cached factor, the projection onto q_sqrt, and the two fused model drivers
// (GPR.log_marginal_likelihood, one shard of SVGP.elbo).
//
// Trapezoidal Cholesky.  A is [(n + extra) x n]: the top square block is factored, the `extra` rows
// below ride along through every panel solve and trailing update and come out as  B L^-T  -- the
// tf.linalg.triangular_solve of the reference fused into the factorisation.  Two-level right-looking:
//   outer panels of NBO = 512 columns  -> trailing update is a K = 512 MFMA GEMM (64 flop/B on C),
//   inner blocks of NB = 128 columns   -> leaf kernel (L11 and L11^-1), in-place panel solve
//                                          A21 <- A21 * L11^-T as a GEMM, update of the rest of the panel.
#include "gpk_internal.h"
#include <stdlib.h>

namespace {
constexpr int NB = GPK_NB;
constexpr int NBO = 512;

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
}  // namespace

extern "C" const char* gpk_version(void) { return "gpk 0.1 (gfx950, fp64 MFMA)"; }

extern "C" size_t gpk_invd_elems(int n, int batch) {
  return (size_t)(batch > 0 ? batch : 1) * gpk_cdiv(n, NB) * NB * NB;
}

// ---- auxiliary streams + event pool (one set per device, created lazily) --------------------------
// The factorisation runs on two streams of its own, forked from / joined to the caller's stream with
// events only (so the whole sequence stays hipGraph-capturable):
//   P  "panel" stream, high priority, all CUs: the latency-bound critical path (leaf, panel solve, inner
//      updates) of the NEXT outer panel (look-ahead);
//   B  "bulk" stream: the big MFMA GEMMs of the outer trailing updates.  For n >= 4096 it is CU-masked;
//   X  four "extra rows" streams: the right-looking solve of the extra rows (SVGP minibatch), one row
//      quarter per stream (the in-group steps are short dependent GEMMs: four of them in flight fill the
//      machine), which nothing on the critical path waits for; confined to the upper 5/8 of the CUs so
//      that the latency chain of an SVGP-sized factorisation (P and B) always finds free CUs.
//      B's mask leaves GPK_RESERVED_CUS compute units (default 16 = 2 per XCD; mask bit i is CU
//      i/8 of XCD i%8 on MI355X, tools/cumask_test.hip) to the panel stream: without that the one-workgroup
//      leaf kernel, which needs a whole CU's LDS, queues behind thousands of resident GEMM workgroups
//      (a 2 ms stall per panel at N = 16384 in the rocprof trace) and the look-ahead never overlaps.
// Not re-entrant across host threads for one device (one event pool).
namespace {
struct Aux {
  hipStream_t P = nullptr;   // panel stream, high priority, all CUs
  hipStream_t B = nullptr;   // bulk stream for large n: every CU except the reserved ones
  hipStream_t Bs = nullptr;  // bulk stream for small n: all CUs (its GEMMs are on the critical path there)
  hipStream_t Bl = nullptr;  // bulk stream for the chain-bound tail of a large factorisation: leaves half the CUs to P
  hipStream_t X[4] = {nullptr, nullptr, nullptr, nullptr};  // extra rows: in-group steps (short dependent GEMMs), unmasked
  hipStream_t Xb = nullptr;  // extra rows: the big right-looking updates, CU-masked (leaves GPK_EXTRA_RESERVED_CUS free)
  hipEvent_t* ev = nullptr;
  int nev = 0;
  int* chain_flags = nullptr;  // device ints of the persistent chain kernel (leaf.hip)
  int bulk_cus = 0;            // CUs the masked bulk stream B may use
};
Aux g_aux[16];

int masked_stream(hipStream_t* out, int ncu, int first, int last) {  // CUs [first, last)
  if (first <= 0 && last >= ncu) return (int)hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  uint32_t mask[32] = {0};
  for (int i = first; i < last; ++i) mask[i >> 5] |= 1u << (i & 31);
  return (int)hipExtStreamCreateWithCUMask(out, (uint32_t)((ncu + 31) / 32), mask);
}

int aux_get(int need, Aux** out) {
  int dev = 0;
  GPK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return GPK_E_UNSUPPORTED;
  Aux& a = g_aux[dev];
  if (!a.P) {
    int lo = 0, hi = 0;
    GPK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    static const bool p_normal = getenv("GPK_P_NORMAL") != nullptr;  // A/B knob: panel stream without priority
    GPK_HIP(hipStreamCreateWithPriority(&a.P, hipStreamNonBlocking, p_normal ? lo : hi));
    hipDeviceProp_t prop;
    GPK_HIP(hipGetDeviceProperties(&prop, dev));
    const int ncu = prop.multiProcessorCount;
    int reserved = 16, xfrom = 0;  // X unmasked by default: CU-masked queues dispatched the short extra-row GEMMs slower (A/B: 385 vs 305 steps/s)
    if (const char* e = getenv("GPK_RESERVED_CUS")) reserved = atoi(e);
    if (const char* e = getenv("GPK_EXTRA_CUS_FROM")) xfrom = atoi(e);
    if (ncu > 1024 || reserved < 0 || reserved >= ncu) reserved = 0;
    if (ncu > 1024 || xfrom < 0 || xfrom >= ncu) xfrom = 0;
    int rc = masked_stream(&a.B, ncu, reserved, ncu);
    if (rc) return rc;
    a.bulk_cus = ncu - reserved;
    for (int i = 0; i < 4; ++i) {
      rc = masked_stream(&a.X[i], ncu, xfrom, ncu);
      if (rc) return rc;
    }
    // optional: a CU-masked stream for the one big GEMM per group (A/B on MI355X: the two event hops per group
    // cost more than the leaf stalls they avoid -- 363 vs 385 steps/s -- so it is off unless requested)
    if (const char* e = getenv("GPK_EXTRA_RESERVED_CUS")) {
      int xres = atoi(e);
      if (ncu > 1024 || xres < 0 || xres >= ncu) xres = 0;
      rc = masked_stream(&a.Xb, ncu, xres, ncu);
      if (rc) return rc;
    }
    GPK_HIP(hipStreamCreateWithFlags(&a.Bs, hipStreamNonBlocking));
    GPK_HIP(hipMalloc((void**)&a.chain_flags, gpk_chain_flag_bytes()));  // one-time, 768 bytes of sync words
    int late_res = ncu / 2;
    if (const char* e = getenv("GPK_LATE_RESERVED_CUS")) late_res = atoi(e);
    if (ncu > 1024 || late_res < 0 || late_res >= ncu) late_res = 0;
    rc = masked_stream(&a.Bl, ncu, late_res, ncu);
    if (rc) return rc;
  }
  if (a.nev < need) {
    hipEvent_t* n = (hipEvent_t*)realloc(a.ev, sizeof(hipEvent_t) * need);
    if (!n) return GPK_E_ARG;
    a.ev = n;
    // (hipEventDisableSystemFence measured SLOWER here: 283 vs 308 steps/s on the SVGP step)
    static const unsigned ev_flags =
        getenv("GPK_EVENT_NO_SYSTEM_FENCE") ? (hipEventDisableTiming | hipEventDisableSystemFence) : hipEventDisableTiming;
    for (int i = a.nev; i < need; ++i) GPK_HIP(hipEventCreateWithFlags(&a.ev[i], ev_flags));
    a.nev = need;
  }
  *out = &a;
  return 0;
}

// column offset of a recursive tail factorisation (pivot-failure columns are reported in the caller's numbering) and
// "info already initialised" marker; both only ever set around the one recursive call below
int g_col_base = 0;
bool g_keep_info = false;

// factor the outer panel [c0,c1) of the square part (rows up to `rows`) on stream s
int factor_panel(hipStream_t s, double* A, int rows, int c0, int c1, long lda, int batch, long strideA,
                 double* invd, long strideInv, int* info) {
  int rc;
  for (int j0 = c0; j0 < c1; j0 += NB) {
    const int j1 = (j0 + NB < c1) ? j0 + NB : c1;
    const int nb = j1 - j0;
    double* invb = invd + (long)(j0 / NB) * NB * NB;
    rc = gpk_launch_leaf(s, A + (long)j0 * lda + j0, lda, strideA, nb, invb, strideInv, info, j0 + g_col_base, batch, 0);
    if (rc) return rc;
    const int below = rows - j1;
    if (below <= 0) continue;
    double* panel = A + (long)j1 * lda + j0;
    // in-place panel solve X = panel * inv(L11)^T: one column tile, so each workgroup only
    // overwrites rows that it alone has read
    GemmArgs g = gemm_base(below, nb, nb, 1.0, panel, lda, invb, NB, 0.0, panel, lda, batch, strideA,
                           strideInv, strideA);
    g.b_tri = 2;
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

// extra rows E = A[n:n+extra, :] against the finished outer panel [c0,c1), right-looking:
//   E[:,c0:c1] <- E[:,c0:c1] L[c0:c1,c0:c1]^-T            (NB-blocked, diagonal-block inverses)
//   E[:,c1:n]  -= E[:,c0:c1] L[c1:n,c0:c1]^T               (one large GEMM, K = c1 - c0)
int extra_panel(hipStream_t s, hipStream_t sbig, hipEvent_t ev_in, hipEvent_t ev_big, double* A, int n, int row0,
                int extra, int c0, int c1, long lda, int batch, long strideA, const double* invd, long strideInv,
                double* Eout, long ldeout, hipEvent_t ev_solved = nullptr) {
  double* E = A + (long)(n + row0) * lda;  // rows [row0, row0 + extra) of the extra block
  // solved panel: in place, or (gpk_potrf_ex) in the separate output matrix
  double* So = Eout ? Eout + (long)row0 * ldeout : E;
  const long ldso = Eout ? ldeout : lda;
  int rc;
  // Fused one-launch group solve (trsm.hip): correct, but measured SLOWER than the seven short GEMMs on the SVGP
  // step (354 vs 382 steps/s): 512 workgroups re-read the same 128 KB operand blocks from L2 and occupy every
  // CU while the leaf kernel waits.  Opt-in until it stages its B operands through LDS.
  static const bool use_group = getenv("GPK_TRSM_GROUP") != nullptr;
  if (use_group && batch == 1 && (c1 - c0) % NB == 0 && (c1 - c0) <= 4 * NB) {
    // the whole group in one launch (trsm.hip): rows are independent, 16 per workgroup
    rc = gpk_launch_trsm_group(s, E + c0, lda, So + c0, ldso, extra, A + (long)c0 * lda + c0, lda,
                               invd + (long)(c0 / NB) * NB * NB, (c1 - c0) / NB);
    if (rc) return rc;
  } else {
    // right-looking inside the group too: after block j is solved, ALL remaining columns of the group are updated
    // by one K = 128 GEMM (192 / 128 / 64 tiles) -- the left-looking form (K = 128, 256, 384 on 64 tiles each) was
    // latency-bound at 44 / 58 / 74 us per launch
    for (int j0 = c0; j0 < c1; j0 += NB) {
      const int j1 = (j0 + NB < c1) ? j0 + NB : c1;
      const int nb = j1 - j0;
      GemmArgs g = gemm_base(extra, nb, nb, 1.0, E + j0, lda, invd + (long)(j0 / NB) * NB * NB, NB, 0.0,
                             E + j0, lda, batch, strideA, strideInv, strideA);
      g.b_tri = 2;
      rc = gpk_launch_gemm(s, g);
      if (rc) return rc;
      if (j1 < c1) {
        GemmArgs u = gemm_base(extra, c1 - j1, nb, -1.0, E + j0, lda, A + (long)j1 * lda + j0, lda, 1.0,
                               E + j1, lda, batch, strideA, strideA, strideA);
        rc = gpk_launch_gemm(s, u);
        if (rc) return rc;
      }
    }
    if (Eout) {  // (only reached with batch == 1 and a ragged last group)
      GPK_HIP(hipMemcpy2DAsync(So + c0, ldso * sizeof(double), E + c0, lda * sizeof(double),
                               (size_t)(c1 - c0) * sizeof(double), (size_t)extra, hipMemcpyDeviceToDevice, s));
    }
  }
  if (ev_solved) GPK_HIP(hipEventRecord(ev_solved, s));  // columns [c0,c1) of the extra rows are final
  if (c1 < n) {
    // the one large GEMM of the group: E[:, c1:n] -= S[:, c0:c1] L[c1:n, c0:c1]^T
    if (sbig != s) {
      GPK_HIP(hipEventRecord(ev_in, s));
      GPK_HIP(hipStreamWaitEvent(sbig, ev_in, 0));
    }
    GemmArgs u = gemm_base(extra, n - c1, c1 - c0, -1.0, So + c0, ldso, A + (long)c1 * lda + c0, lda, 1.0,
                           E + c1, lda, batch, Eout ? 0 : strideA, strideA, strideA);
    // optional cap on (persistent) workgroups so that some CUs stay free for the panel stream's leaf kernel
    // (A/B on the SVGP step: cap 320 -> 448 steps/s, no cap 435, cap 224 -> 431)
    static const int xwgs = getenv("GPK_EXTRA_MAX_WGS") ? atoi(getenv("GPK_EXTRA_MAX_WGS")) : 320;
    u.max_wgs = xwgs;
    rc = gpk_launch_gemm(sbig, u);
    if (rc) return rc;
    if (sbig != s) {
      GPK_HIP(hipEventRecord(ev_big, sbig));
      GPK_HIP(hipStreamWaitEvent(s, ev_big, 0));
    }
  }
  return 0;
}

// ---- streamed projection (SVGP step) -----------------------------------------------------------------------------
// LTA = tril(q_sqrt)^T A needs, for output row i, the rows k >= i of A = Lm^-1 Kuf -- the LAST rows the factorisation
// produces -- so as one GEMM it can only start when everything else is over (0.55 ms of the 2.2 ms step, alone on the
// chip).  Right-looking instead: as soon as the extra-row solve has finished columns [g0,g1) of A^T, their
// contribution to ALL outputs i < g1 is added,
//     C[b, i] (+)= sum_{k in [max(i,g0), g1)} At[b,k] LqT[i,k]        (rectangle i < g0: beta = 1; triangle: beta = 0)
// on the extra-row stream, which otherwise idles until the panel chain delivers the next group.  The last group
// runs with the row-sum-of-squares epilogue (C + A B^T is squared, not stored).  Set by gpk_svgp_elbo_shard around
// its factorisation call only.
struct ProjStream {
  bool on = false;
  const double* LqT = nullptr; long ldl = 0, strideL = 0;   // [P][m, ldl]  LqT[i,k] = q_sqrt[k,i], zero for k < i
  double* C = nullptr; long ldc = 0, strideC = 0;           // [P][rows, ldc] running A^T Lq
  double* part = nullptr; long part_ld = 0, stridePart = 0; // [P][2 * tiles(m), rows] partial row sums of squares
  int P = 0;
  int groups = 0;                                           // groups issued (0 afterwards = the hook never ran)
};
ProjStream g_proj;

int proj_group(hipStream_t s, const double* At, long ldat, int row0, int nrows, int g0, int g1, int m) {
  ProjStream& q = g_proj;
  const bool last = g1 == m;
  // while the panel chain is still running these GEMMs are capped like the big extra-row update (persistent
  // workgroups), so that the leaf kernel finds a free CU
  static const int pwgs = getenv("GPK_PROJ_MAX_WGS") ? atoi(getenv("GPK_PROJ_MAX_WGS")) : 320;
  const double* Ag = At + (long)row0 * ldat + g0;
  double* Cg = q.C + (long)row0 * q.ldc;
  int rc;
  if (g0 > 0) {
    GemmArgs r = gemm_base(nrows, g0, g1 - g0, 1.0, Ag, ldat, q.LqT + g0, q.ldl, 1.0, Cg, q.ldc, q.P, 0, q.strideL,
                           q.strideC);
    if (!last) r.max_wgs = pwgs;
    if (last) {
      r.epi = 1; r.sq_cols = g0; r.c2_cols = 0;
      r.part = q.part + row0; r.part_ld = q.part_ld; r.stridePart = q.stridePart;
      r.C2 = q.part; r.ldc2 = 0; r.strideC2 = 0;
    }
    rc = gpk_launch_gemm(s, r);
    if (rc) return rc;
  }
  GemmArgs t = gemm_base(nrows, g1 - g0, g1 - g0, 1.0, Ag, ldat, q.LqT + (long)g0 * q.ldl + g0, q.ldl, 0.0, Cg + g0,
                         q.ldc, q.P, 0, q.strideL, q.strideC);
  t.b_tri = 1;
  if (!last) t.max_wgs = pwgs;
  if (last) {
    t.C = nullptr;
    t.epi = 1; t.sq_cols = g1 - g0; t.c2_cols = 0;
    t.part = q.part + (long)(g0 / NB) * 2 * q.part_ld + row0; t.part_ld = q.part_ld; t.stridePart = q.stridePart;
    t.C2 = q.part; t.ldc2 = 0; t.strideC2 = 0;
  }
  rc = gpk_launch_gemm(s, t);
  if (rc) return rc;
  ++q.groups;
  return 0;
}
}  // namespace

namespace {

int potrf_core(hipStream_t S, double* A, int n, int extra, long lda, int batch, long strideA, double* invd, int zero_upper,
               int* info, double* Eout, long ldeout, double* ws);
}  // namespace

// ---- hipGraph replay of an SVGP-sized factorisation ------------------------------------------------------------
// The chain  leaf -> solve -> strip  of a 2048-column factorisation is ~50 launches and ~60 event operations on
// three streams, and at 7-40 us per kernel the HOST issue rate, not the GPU, paced it (removing the device-side
// event waits changed nothing: A/B).  The enqueue sequence depends only on the arguments, so the second call with the
// same arguments captures it (stream capture follows the event fork/join onto the internal streams) and later
// calls replay the instantiated graph: one hipGraphLaunch per factorisation.
namespace {
struct GraphKey {
  void* stream; double* A; int n, extra; long lda; int batch; long strideA; double* invd; int zero_upper; int* info;
  double* Eout; long ldeout;
  bool operator==(const GraphKey& o) const {
    return stream == o.stream && A == o.A && n == o.n && extra == o.extra && lda == o.lda && batch == o.batch &&
           strideA == o.strideA && invd == o.invd && zero_upper == o.zero_upper && info == o.info && Eout == o.Eout &&
           ldeout == o.ldeout;
  }
};
struct GraphEntry { GraphKey key; int seen; hipGraphExec_t exec; };
GraphEntry g_graphs[8];
int g_ngraphs = 0, g_graph_next = 0;

int potrf_maybe_graph(const GraphKey& k) {
  // Opt-in (GPK_GRAPH=1).  Measured on MI355X / ROCm 7.2: replay is SLOWER than eager issue (339 vs 370 steps/s on a
  // created stream; eager on the default stream reaches 435-450), and the HIP 7.0 runtime bundled with PyTorch
  // overflows its stack in hipStreamEndCapture on this cross-linked stream topology (infinite recursion).
  static const bool enabled = getenv("GPK_GRAPH") != nullptr;
  hipStream_t S = (hipStream_t)k.stream;
  // (the legacy default stream cannot be captured: callers on stream 0 get the eager path)
  const bool eligible = enabled && S != nullptr && k.batch <= 1 && k.n > NB && k.n < 4096 && k.extra > 256 &&
                        !gpk_profile_gemm_is_on();
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (eligible && hipStreamIsCapturing(S, &cs) != hipSuccess) cs = hipStreamCaptureStatusActive;
  if (!eligible || cs != hipStreamCaptureStatusNone)
    return potrf_core(S, k.A, k.n, k.extra, k.lda, k.batch, k.strideA, k.invd, k.zero_upper, k.info, k.Eout, k.ldeout, nullptr);
  GraphEntry* e = nullptr;
  for (int i = 0; i < g_ngraphs; ++i)
    if (g_graphs[i].key == k) { e = &g_graphs[i]; break; }
  if (e && e->exec) {
    const hipError_t le = hipGraphLaunch(e->exec, S);
    if (le != hipSuccess && getenv("GPK_DEBUG")) fprintf(stderr, "[gpk] hipGraphLaunch (replay) -> %d\n", (int)le);
    return (int)le;
  }
  if (!e) {  // first sighting: run eagerly (also performs the one-time kernel attribute / stream / event setup)
    if (g_ngraphs < 8) e = &g_graphs[g_ngraphs++];
    else {
      e = &g_graphs[g_graph_next];
      g_graph_next = (g_graph_next + 1) % 8;
      if (e->exec) (void)hipGraphExecDestroy(e->exec);
    }
    e->key = k; e->seen = 1; e->exec = nullptr;
    return potrf_core(S, k.A, k.n, k.extra, k.lda, k.batch, k.strideA, k.invd, k.zero_upper, k.info, k.Eout, k.ldeout, nullptr);
  }
  // second sighting: capture, instantiate, launch
  hipGraph_t graph = nullptr;
  if (getenv("GPK_DEBUG")) fprintf(stderr, "[gpk] begin capture on %p\n", (void*)S);
  if (hipStreamBeginCapture(S, hipStreamCaptureModeRelaxed) != hipSuccess)
    return potrf_core(S, k.A, k.n, k.extra, k.lda, k.batch, k.strideA, k.invd, k.zero_upper, k.info, k.Eout, k.ldeout, nullptr);
  const int rc = potrf_core(S, k.A, k.n, k.extra, k.lda, k.batch, k.strideA, k.invd, k.zero_upper, k.info, k.Eout, k.ldeout, nullptr);
  const hipError_t ce = hipStreamEndCapture(S, &graph);
  if (getenv("GPK_DEBUG")) fprintf(stderr, "[gpk] capture: enqueue rc %d, end-capture %d, graph %p\n", rc, (int)ce, (void*)graph);
  if (rc != 0 || ce != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    e->seen = -1000000;  // never try again for this key
    if (rc != 0) return rc;
    return potrf_core(S, k.A, k.n, k.extra, k.lda, k.batch, k.strideA, k.invd, k.zero_upper, k.info, k.Eout, k.ldeout, nullptr);
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (getenv("GPK_DEBUG")) fprintf(stderr, "[gpk] instantiate -> %d\n", (int)ie);
  if (ie != hipSuccess || !exec) {
    (void)hipGetLastError();
    return potrf_core(S, k.A, k.n, k.extra, k.lda, k.batch, k.strideA, k.invd, k.zero_upper, k.info, k.Eout, k.ldeout, nullptr);
  }
  e->exec = exec;
  const hipError_t le = hipGraphLaunch(exec, S);
  if (le != hipSuccess && getenv("GPK_DEBUG")) fprintf(stderr, "[gpk] hipGraphLaunch (first) -> %d\n", (int)le);
  return (int)le;
}
}  // namespace

extern "C" int gpk_potrf(void* stream, double* A, int n, int extra, long lda, int batch,
                         long strideA, double* invd, int zero_upper, int* info) {
  if (!A || !invd || n < 0 || extra < 0 || lda < n) return GPK_E_ARG;
  GraphKey k{stream, A, n, extra, lda, batch, strideA, invd, zero_upper, info, nullptr, 0};
  return potrf_maybe_graph(k);
}

extern "C" size_t gpk_potrf_ex_workspace_bytes(void) { return 0; }

// Same factorisation, but the solved extra rows  B L^-T  are written to Eout [extra, n] (leading dimension
// ldeout) instead of in place (the extra rows of A are consumed as scratch).  batch must be 1.
extern "C" int gpk_potrf_ex(void* stream, double* A, int n, int extra, long lda, int batch, long strideA, double* invd,
                            int zero_upper, int* info, double* Eout, long ldeout, void* ws, size_t ws_bytes) {
  (void)ws; (void)ws_bytes;
  if (!Eout || ldeout < n || batch > 1) return GPK_E_ARG;
  if (extra > 256 && n > NB)  // the only shape in which the extra rows are solved apart from the square part
    return potrf_core((hipStream_t)stream, A, n, extra, lda, 1, strideA, invd, zero_upper, info, Eout, ldeout, nullptr);
  int rc = potrf_core((hipStream_t)stream, A, n, extra, lda, 1, strideA, invd, zero_upper, info, nullptr, 0, nullptr);
  if (rc || extra == 0) return rc;
  GPK_HIP(hipMemcpy2DAsync(Eout, ldeout * sizeof(double), A + (long)n * lda, lda * sizeof(double),
                           (size_t)n * sizeof(double), (size_t)extra, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

namespace {
int potrf_core(hipStream_t S, double* A, int n, int extra, long lda, int batch, long strideA, double* invd, int zero_upper,
               int* info, double* Eout, long ldeout, double* ws) {
  if (!A || !invd || n < 0 || extra < 0 || lda < n) return GPK_E_ARG;
  if (batch <= 0) batch = 1;
  if (info && !g_keep_info) GPK_HIP(hipMemsetAsync(info, 0, sizeof(int) * batch, S));
  if (n == 0) return 0;
  const long strideInv = (long)gpk_cdiv(n, NB) * NB * NB;
  // outer panel width: 512 for the large GPR factorisations (K = 512 trailing GEMMs), one leaf block for
  // the SVGP-sized ones, where the whole factorisation is a latency chain of leaf -> solve -> strip
  // outer panel width for n >= 4096: A/B at N = 16384 (same box): 384 -> 34.6 ms, 512 -> 33.2, 640 -> 32.6, 768 -> 32.4,
  // 896 -> 32.3, 1024 -> 32.7
  static const int nbo_large = getenv("GPK_NBO") ? (atoi(getenv("GPK_NBO")) / NB) * NB : 768;
  const int nbo = (n >= 4096) ? (nbo_large >= NB ? nbo_large : NBO) : NB;
  const int npanels = gpk_cdiv(n, nbo);
  // Few extra rows (GPR: the P columns of Y) simply ride along through the panel solves and trailing
  // updates of the square part; many extra rows (SVGP: the minibatch) are solved right-looking on their
  // own stream (extra_panel), overlapped with the factorisation.
  const bool ride = extra > 0 && extra <= 256;
  const int R = ride ? n + extra : n;  // rows handled together with the square part
  int rc;
  if (n <= NB) {  // one leaf; nothing to overlap
    rc = factor_panel(S, A, R, 0, n, lda, batch, strideA, invd, strideInv, info);
    if (rc) return rc;
    if (extra > 0 && !ride) {
      rc = extra_panel(S, S, nullptr, nullptr, A, n, 0, extra, 0, n, lda, batch, strideA, invd, strideInv, Eout, ldeout);
      if (rc) return rc;
    }
    return zero_upper ? gpk_launch_zero_upper(S, A, n, lda, batch, strideA) : 0;
  }
  Aux* aux = nullptr;
  rc = aux_get(2 * npanels + 16, &aux);
  if (rc) return rc;
  hipStream_t P = aux->P, B = (n >= 4096) ? aux->B : aux->Bs;
  // GPK_MERGE_BULK: the extra-row work shares the bulk stream (rest-updates and extra-row GEMMs then execute in
  // issue order instead of competing for CUs)
  static const bool merge_bulk = getenv("GPK_MERGE_BULK") != nullptr;
  hipStream_t Xq[4] = {aux->X[0], aux->X[1], aux->X[2], aux->X[3]};
  if (merge_bulk) Xq[0] = B;
  // extra rows in up to 4 chunks of whole 128-row tiles, one stream each
  int nx = 0, xrow[5] = {0, 0, 0, 0, 0};
  if (extra > 0 && !ride) {
    // more streams than hardware queues serialise against each other (measured: 4 were slower than 1)
    static const int nx_env = getenv("GPK_EXTRA_STREAMS") ? atoi(getenv("GPK_EXTRA_STREAMS")) : 1;
    nx = nx_env < 1 ? 1 : (nx_env > 4 ? 4 : nx_env);
    while (nx > 1 && extra < nx * 512) --nx;
    const int per = gpk_cdiv(gpk_cdiv(extra, nx), NB) * NB;
    for (int i = 0; i <= nx; ++i) xrow[i] = (i * per < extra) ? i * per : extra;
  }
  hipEvent_t* evF = aux->ev;            // [npanels] panel p factored, rows below solved (recorded on P)
  hipEvent_t* evR = aux->ev + npanels;  // [npanels] rest of the trailing update of panel p done (on B)
  hipEvent_t evFork = aux->ev[2 * npanels], evJoinP = aux->ev[2 * npanels + 1],
             evJoinB = aux->ev[2 * npanels + 2];
  const bool useX = extra > 0 && !ride;
  if (getenv("GPK_DEBUG")) fprintf(stderr, "[gpk] potrf_core: fork\n");
  GPK_HIP(hipEventRecord(evFork, S));  // fork: everything already queued on S comes first
  GPK_HIP(hipStreamWaitEvent(P, evFork, 0));
  GPK_HIP(hipStreamWaitEvent(B, evFork, 0));
  for (int i = 0; i < nx; ++i) GPK_HIP(hipStreamWaitEvent(Xq[i], evFork, 0));
  // ---- SVGP-sized factorisations: the whole latency chain is ONE persistent kernel (leaf.hip) -----------------------
  // Opt-in (GPK_CHAIN=1), and only with GPU_MAX_HW_QUEUES >= 4: the leaf launch, the row-owner launch and the two
  // bulk streams (blocked in hipStreamWaitValue32 most of the time) must all be resident hardware queues at once -- with
  // 2 hardware queues a value-wait occupies a queue for good and the owners' launch is never scheduled (bench.py hung).
  // Measured 364 steps/s (4 queues) against 440 for the per-step kernels with 2 queues, so the default stays per-step.
  static const bool no_chain = getenv("GPK_CHAIN") == nullptr;
  if (!no_chain && batch == 1 && n % NB == 0 && n >= 2 * NB && n <= 2048 && !ride && aux->chain_flags) {
    const int np = n / NB;
    int* flags = aux->chain_flags;
    int* fPP = flags + gpk_chain_flag_index(0);
    int* fRB = flags + gpk_chain_flag_index(1);
    hipEvent_t evZ = aux->ev[2 * npanels + 9];
    hipStream_t W = aux->X[2];  // the row-block owners' launch (plain stream, otherwise unused)
    GPK_HIP(hipMemsetAsync(flags, 0, gpk_chain_flag_bytes(), P));
    GPK_HIP(hipEventRecord(evZ, P));
    GPK_HIP(hipStreamWaitEvent(W, evZ, 0));
    rc = gpk_launch_chain(P, W, A, lda, n, invd, info, flags);
    if (rc) return rc;
    GPK_HIP(hipEventRecord(aux->ev[2 * npanels + 10], W));
    GPK_HIP(hipStreamWaitEvent(S, aux->ev[2 * npanels + 10], 0));
    GPK_HIP(hipStreamWaitEvent(B, evZ, 0));
    hipStream_t Xs = Xq[0];
    if (useX) GPK_HIP(hipStreamWaitEvent(Xs, evZ, 0));
    int g0 = 0;
    // (stream value-waits are issued in the order the chain will satisfy them: the host call can block until an
    //  earlier wait of the same hardware queue has been consumed)
    for (int p = 0; p < np; ++p) {
      const int c0 = p * NB, c1 = c0 + NB, c2 = c0 + 2 * NB;
      if (c2 < n) {  // rest-update of panel p: columns from panel p + 2 on, K = 128
        GPK_HIP(hipStreamWaitValue32(B, fPP, (uint32_t)(p + 1), hipStreamWaitValueGte, 0xffffffffu));
        const double* P2 = A + (long)c2 * lda + c0;
        GemmArgs u = gemm_base(n - c2, n - c2, NB, -1.0, P2, lda, P2, lda, 1.0, A + (long)c2 * lda + c2, lda, 1, 0, 0, 0);
        u.c_lower = 1;
        rc = gpk_launch_gemm(B, u);
        if (rc) return rc;
        GPK_HIP(hipStreamWriteValue32(B, fRB, (uint32_t)(p + 1), 0));
      }
      if (useX) {
        // groups of 512 columns while the chain is running; the LAST 512 columns (nothing left to overlap with:
        // the chain has finished by then) go through the fused one-launch group solve (trsm.hip) instead of ten
        // short dependent launches.  GPK_CHAIN_TAIL=0 restores the shrinking tail groups.
        static const bool fused_tail = !(getenv("GPK_CHAIN_TAIL") && atoi(getenv("GPK_CHAIN_TAIL")) == 0);
        const bool use_fused = fused_tail && (n % NBO) == 0 && n >= 2 * NBO;
        const bool tail_group = !use_fused && (n >= 8 * NB) && (c1 == n - 2 * NB || c1 == n - NB);
        const bool full_group = (c1 % NBO) == 0 && !(!use_fused && (n >= 8 * NB) && c1 > n - 2 * NB && c1 < n);
        if (use_fused && c1 == n) {
          GPK_HIP(hipStreamWaitValue32(Xs, fPP, (uint32_t)np, hipStreamWaitValueGte, 0xffffffffu));
          double* E = A + (long)n * lda;
          double* So = Eout ? Eout : E;
          const long ldso = Eout ? ldeout : lda;
          rc = gpk_launch_trsm_group(Xs, E + g0, lda, So + g0, ldso, extra, A + (long)g0 * lda + g0, lda,
                                     invd + (long)(g0 / NB) * NB * NB, (n - g0) / NB);
          if (rc) return rc;
          g0 = c1;
        } else if ((c1 == n && !use_fused) || (full_group && c1 < n) || tail_group) {
          GPK_HIP(hipStreamWaitValue32(Xs, fPP, (uint32_t)(c1 / NB), hipStreamWaitValueGte, 0xffffffffu));
          rc = extra_panel(Xs, Xs, nullptr, nullptr, A, n, 0, extra, g0, c1, lda, 1, strideA, invd, strideInv, Eout, ldeout);
          if (rc) return rc;
          g0 = c1;
        }
      }
    }
    if (useX) {
      GPK_HIP(hipEventRecord(aux->ev[2 * npanels + 3], Xs));
      GPK_HIP(hipStreamWaitEvent(S, aux->ev[2 * npanels + 3], 0));
    }
    GPK_HIP(hipEventRecord(evJoinP, P));
    GPK_HIP(hipEventRecord(evJoinB, B));
    GPK_HIP(hipStreamWaitEvent(S, evJoinP, 0));
    GPK_HIP(hipStreamWaitEvent(S, evJoinB, 0));
    if (zero_upper) return gpk_launch_zero_upper(S, A, n, lda, 1, strideA);
    return 0;
  }
  hipStream_t last_bulk = B;
  // Deferred rest-updates: while the trailing matrix is large, the far trailing update is applied once per TWO
  // outer panels, as a K = 1024 GEMM (65 vs 59 TFLOP/s for K = 512 on this chip); the strip of the look-ahead
  // carries whatever panels are still pending for the next panel's columns.  r0 = first pending column.
  bool proj_side_used = false;
  int r0 = 0, last_rest = -1;  // last_rest: panel index whose evR marks the most recent rest-update
  int xg0 = 0;                 // first column of the current extra-row group
  static const int defer_rows = getenv("GPK_DEFER_ROWS") ? atoi(getenv("GPK_DEFER_ROWS")) : (1 << 30);  // off by default: A/B 33.7 vs 32.7 ms at N = 16384 (the K = 1024 strips starve on the panel stream)
  for (int p = 0; p < npanels; ++p) {
    const int c0 = p * nbo;
    const int c1 = (c0 + nbo < n) ? c0 + nbo : n;
    const int c2 = (c1 + nbo < n) ? c1 + nbo : n;
    // ---- P: the critical path.  Panel p, then the strip = columns of panel p+1 (look-ahead) -----------
    rc = factor_panel(P, A, R, c0, c1, lda, batch, strideA, invd, strideInv, info);
    if (rc) return rc;
    static const bool dbg_late_record = getenv("GPK_DBG_LATE_RECORD") != nullptr;   // timing experiments only
    static const bool dbg_no_wait = getenv("GPK_DBG_NO_WAIT") != nullptr;           // (races: wrong results)
    if (!dbg_late_record) GPK_HIP(hipEventRecord(evF[p], P));
    const int kpend = c1 - r0;                   // pending columns r0:c1 (one or two panels)
    const double* Pn = A + (long)c1 * lda + r0;  // rows c1.. of the pending panels (solved)
    if (c1 < n) {
      // columns c1:c2 also received the most recent rest-update (on a bulk stream): order the two
      if (last_rest >= 0 && !dbg_no_wait) GPK_HIP(hipStreamWaitEvent(P, evR[last_rest], 0));
      GemmArgs u = gemm_base(R - c1, c2 - c1, kpend, -1.0, Pn, lda, Pn, lda, 1.0,
                             A + (long)c1 * lda + c1, lda, batch, strideA, strideA, strideA);
      u.c_lower = 1;
      rc = gpk_launch_gemm(P, u);
      if (rc) return rc;
    }
    if (dbg_late_record) GPK_HIP(hipEventRecord(evF[p], P));
    // ---- B: rest of the outer trailing update  A[c2:, c2:] -= P[c2:] P[c2:]^T, lower tiles only --------
    // While the trailing matrix is large the factorisation is bound by these GEMMs and they start as soon as
    // panel p is solved.  Near the end it is bound by the latency chain of P instead: there the strip goes
    // first (alone on the chip) and the rest-update overlaps the NEXT panel's chain rather than the strip.
    static const int late_rows = getenv("GPK_LATE_ROWS") ? atoi(getenv("GPK_LATE_ROWS")) : 3072;  // A/B at N = 16384: 3072 -> 32.9 ms, 6144 -> 33.8, off -> 33.1
    const bool strip_first = (n >= 4096) && (n - c1 <= late_rows) && (c1 < n);
    const bool defer = (nbo > NB) && (kpend < 2 * nbo) && (n - c2 >= defer_rows) && (c2 + nbo < n);
    if (c2 < n && !defer) {
      hipStream_t Bp = B;
      if (strip_first) {
        Bp = aux->Bl;  // (in-order with the earlier rest-updates through evR below)
        GPK_HIP(hipEventRecord(aux->ev[2 * npanels + 9], P));
        GPK_HIP(hipStreamWaitEvent(Bp, aux->ev[2 * npanels + 9], 0));
        if (last_rest >= 0) GPK_HIP(hipStreamWaitEvent(Bp, evR[last_rest], 0));
      } else {
        GPK_HIP(hipStreamWaitEvent(B, evF[p], 0));
      }
      const double* P2 = A + (long)c2 * lda + r0;
      GemmArgs u = gemm_base(R - c2, n - c2, kpend, -1.0, P2, lda, P2, lda, 1.0,
                             A + (long)c2 * lda + c2, lda, batch, strideA, strideA, strideA);
      u.c_lower = 1;
      // A/B knob: with many extra rows in flight (SVGP step) the rest-update's one-shot kernel (150 KB LDS per workgroup)
      // only fits on CUs free of extra-row GEMM workgroups; the tiled kernel can share a CU
      static const int rest_no_small = getenv("GPK_REST_NO_SMALL") ? atoi(getenv("GPK_REST_NO_SMALL")) : 0;
      if (rest_no_small && useX && nbo == NB) u.no_small = 1;
      if (Bp == aux->B && n >= 4096) u.stagger_first = aux->bulk_cus;
      rc = gpk_launch_gemm(Bp, u);
      if (rc) return rc;
      GPK_HIP(hipEventRecord(evR[p], Bp));
      last_bulk = Bp;
      last_rest = p;
      r0 = c1;
    } else if (!defer) {
      r0 = c1;
    }
    // ---- X: the extra rows against panel p -----------------------------------------------------------
    // (in groups of up to 512 columns, so that its big right-looking update is a K = 512 GEMM)
    // Groups shrink towards the end (.., n-256, n-128, n): whatever is left of the extra-row solve when the LAST leaf
    // finishes is exposed latency, and with one-block groups that is a single short launch instead of seven.
    const bool tail_group = (nbo == NB) && (n >= 8 * NB) && (c1 == n - 2 * NB || c1 == n - NB);
    static const int xgroup = getenv("GPK_XGROUP") ? (atoi(getenv("GPK_XGROUP")) / NB) * NB : NBO;  // A/B knob
    const bool full_group = (c1 % (xgroup >= NB ? xgroup : NBO)) == 0 && !((nbo == NB) && (n >= 8 * NB) && c1 > n - 2 * NB && c1 < n);
    if (useX && (c1 == n || full_group || tail_group)) {
      const int g0 = xg0;
      xg0 = c1;
      for (int i = 0; i < nx; ++i) {
        if (xrow[i + 1] <= xrow[i]) continue;
        GPK_HIP(hipStreamWaitEvent(Xq[i], evF[p], 0));
        const bool proj = g_proj.on && batch == 1 && !Eout;
        // GPK_PROJ_SIDE=1: the projection GEMMs go to a stream of their own (ordered after the group's solve, concurrent
        // with its big update) instead of following the big update on the extra-row stream
        const bool proj_side = getenv("GPK_PROJ_SIDE") != nullptr;  // (read per call: the tests toggle it)
        hipEvent_t evS = (proj && proj_side && nx == 1) ? aux->ev[2 * npanels + 11] : nullptr;
        rc = extra_panel(Xq[i], aux->Xb ? aux->Xb : Xq[i], aux->ev[2 * npanels + 7], aux->ev[2 * npanels + 8], A, n, xrow[i],
                         xrow[i + 1] - xrow[i], g0, c1, lda, batch, strideA, invd, strideInv, Eout, ldeout, evS);
        if (rc) return rc;
        if (proj) {
          hipStream_t ps = Xq[i];
          if (evS) {
            ps = aux->X[1];
            GPK_HIP(hipStreamWaitEvent(ps, evS, 0));
            proj_side_used = true;
          }
          rc = proj_group(ps, A + (long)n * lda, lda, xrow[i], xrow[i + 1] - xrow[i], g0, c1, n);
          if (rc) return rc;
        }
      }
    }
    // ---- tail of a large factorisation: once <= 2048 columns remain it is a pure latency chain, and the SVGP-sized
    // scheme (128-column panels, one-shot LDS-DMA GEMMs for solve and strip) runs it ~2x faster than 512-column panels
    // (opt-in, GPK_TAIL_RECURSION=1: A/B at N = 16384 gave 33.5 ms with it vs 33.25 ms without)
    static const bool tail_recursion = getenv("GPK_TAIL_RECURSION") != nullptr;
    static const int tail_rows = getenv("GPK_TAIL_ROWS") ? atoi(getenv("GPK_TAIL_ROWS")) : 2048;  // (< 4096: the inner call must take the 128-column scheme)
    if (tail_recursion && nbo > NB && batch == 1 && !useX && c1 < n && n - c1 <= tail_rows && n - c1 < 4096 && n - c1 >= 2 * NBO) {
      if (r0 != c1) continue;  // (a deferred rest-update is still pending: not at a clean boundary)
      GPK_HIP(hipEventRecord(evJoinP, P));
      GPK_HIP(hipEventRecord(evJoinB, last_bulk));
      GPK_HIP(hipStreamWaitEvent(S, evJoinP, 0));
      GPK_HIP(hipStreamWaitEvent(S, evJoinB, 0));
      g_col_base = c1;
      g_keep_info = true;
      rc = potrf_core(S, A + (long)c1 * lda + c1, n - c1, extra, lda, 1, strideA, invd + (long)(c1 / NB) * NB * NB, 0, info,
                      nullptr, 0, nullptr);
      g_col_base = 0;
      g_keep_info = false;
      if (rc) return rc;
      if (zero_upper) return gpk_launch_zero_upper(S, A, n, lda, batch, strideA);
      return 0;
    }

  }
  if (getenv("GPK_DEBUG")) fprintf(stderr, "[gpk] potrf_core: join\n");
  // join: P has waited for every rest-update it depends on; B's last event covers the rest
  GPK_HIP(hipEventRecord(evJoinP, P));
  GPK_HIP(hipEventRecord(evJoinB, last_bulk));  // rest-updates are chained through evR, the last one covers all
  GPK_HIP(hipStreamWaitEvent(S, evJoinP, 0));
  GPK_HIP(hipStreamWaitEvent(S, evJoinB, 0));
  for (int i = 0; i < nx; ++i) {
    hipEvent_t ej = aux->ev[2 * npanels + 3 + i];
    GPK_HIP(hipEventRecord(ej, Xq[i]));
    GPK_HIP(hipStreamWaitEvent(S, ej, 0));
  }
  if (proj_side_used) {
    GPK_HIP(hipEventRecord(aux->ev[2 * npanels + 12], aux->X[1]));
    GPK_HIP(hipStreamWaitEvent(S, aux->ev[2 * npanels + 12], 0));
  }
  if (zero_upper) return gpk_launch_zero_upper(S, A, n, lda, batch, strideA);
  return 0;
}
}  // namespace

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

// trans = 0:  B <- B L^-T  with (L, invd);   trans = 1:  B <- B L^-1 with (LT = L^T, invdT)
extern "C" int gpk_trsm(void* stream, int trans, const double* L, long ldl, const double* invd,
                        int n, double* B, int m, long ldb, int batch, long strideL, long strideB) {
  if (!L || !invd || !B || n < 0 || m < 0) return GPK_E_ARG;
  if (n == 0 || m == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (batch <= 0) batch = 1;
  const int nblk = gpk_cdiv(n, NB);
  const long strideInv = (long)nblk * NB * NB;
  int rc;
  if (trans == 0) {
    for (int jb = 0; jb < nblk; ++jb) {
      const int j0 = jb * NB, j1 = (j0 + NB < n) ? j0 + NB : n, nb = j1 - j0;
      if (j0 > 0) {
        GemmArgs u = gemm_base(m, nb, j0, -1.0, B, ldb, L + (long)j0 * ldl, ldl, 1.0, B + j0, ldb,
                               batch, strideB, strideL, strideB);
        rc = gpk_launch_gemm(s, u);
        if (rc) return rc;
      }
      GemmArgs g = gemm_base(m, nb, nb, 1.0, B + j0, ldb, invd + (long)jb * NB * NB, NB, 0.0, B + j0,
                             ldb, batch, strideB, strideInv, strideB);
      g.b_tri = 2;
      rc = gpk_launch_gemm(s, g);
      if (rc) return rc;
    }
  } else {
    for (int jb = nblk - 1; jb >= 0; --jb) {
      const int j0 = jb * NB, j1 = (j0 + NB < n) ? j0 + NB : n, nb = j1 - j0;
      if (j1 < n) {
        // B[:, j0:j1] -= B[:, j1:n] * (LT[j0:j1, j1:n])^T
        GemmArgs u = gemm_base(m, nb, n - j1, -1.0, B + j1, ldb, L + (long)j0 * ldl + j1, ldl, 1.0,
                               B + j0, ldb, batch, strideB, strideL, strideB);
        rc = gpk_launch_gemm(s, u);
        if (rc) return rc;
      }
      GemmArgs g = gemm_base(m, nb, nb, 1.0, B + j0, ldb, invd + (long)jb * NB * NB, NB, 0.0, B + j0,
                             ldb, batch, strideB, strideInv, strideB);
      g.b_tri = 1;
      rc = gpk_launch_gemm(s, g);
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
#define GPK_STREAM_PROJ_DEFAULT 0
#endif
namespace {
struct ElboLayout {
  long ld; int nt;
  size_t off_T, off_invd, off_LqT, off_s0, off_fmean, off_ssq, off_proj, off_part0, off_part1, off_At, off_pex, off_C, total;
};
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
  l.off_At = o; o += gpk_align_up((size_t)rows * l.ld * sizeof(double), 256);   // A^T = Kfu Lm^-T (gpk_potrf_ex output)
  l.off_pex = o; o += gpk_align_up(gpk_potrf_ex_workspace_bytes(), 256);
  l.off_C = o; o += q_diag ? 0 : gpk_align_up((size_t)P * rows * l.ld * sizeof(double), 256);  // running A^T Lq (streamed projection)
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
  double* Kfu = T + (long)m * l.ld;        // extra rows of the trapezoid: Kfu, consumed by the factorisation
  double* At = (double*)(w + l.off_At);   // A^T = Kfu Lm^-T
  int rc;
  // Kuf^T = k(Xb, Z) as the extra rows (posteriors.py:836, covariances/kufs.py:31-34).  Only the extra-row
  // stream of the factorisation consumes it, so it is built THERE (ordered after everything already queued on
  // the caller's stream) and the panel chain starts right after the much smaller Kuu build.
  static const bool plain_streams = !getenv("GPK_EXTRA_STREAMS") && !getenv("GPK_MERGE_BULK");
  hipStream_t kfu_stream = s;
  // (with hipGraph replay of the factorisation the graph is launched on the caller's stream and cannot depend on
  // eager work of an internal stream: the build then stays on the caller's stream)
  static const bool graphs = getenv("GPK_GRAPH") != nullptr;
  if (plain_streams && !graphs && m > GPK_NB && rows > 256) {
    Aux* aux = nullptr;
    rc = aux_get(8, &aux);
    if (rc) return rc;
    GPK_HIP(hipEventRecord(aux->ev[0], s));  // (event slot 0 is re-recorded by gpk_potrf only after this wait was queued)
    GPK_HIP(hipStreamWaitEvent(aux->X[0], aux->ev[0], 0));
    kfu_stream = aux->X[0];
  }
  rc = gpk_kernel_matrix((void*)kfu_stream, family, Xb, rows, ldxb, Z, m, ldz, d, ls_host, ard, variance, 0.0, 0,
                         Kfu, l.ld);
  if (rc) return rc;
  // Work that depends on neither factorisation nor minibatch solve -- tril(q_sqrt)^T for the projection and the whole
  // KL term -- also goes to that stream, which idles until the first 512 columns of Lm exist; gpk_potrf joins it.
  const bool side = kfu_stream != s;
  int c1 = 0;
  if (side) {
    if (!q_diag) {
      rc = gpk_transpose((void*)kfu_stream, q_sqrt, m, m, m, LqT, l.ld, 1, P, (long)m * m, (long)m * l.ld);
      if (rc) return rc;
    }
    rc = gpk_launch_kl_white_stage1(kfu_stream, q_mu, q_sqrt, m, P, q_diag, part1, &c1);
    if (rc) return rc;
    const double* p1s[1] = {part1};
    const double halfs = 0.5;
    rc = gpk_launch_final(kfu_stream, 1, p1s, &c1, &halfs, -0.5 * (double)m * (double)P, out + 1);
    if (rc) return rc;
  }
  // Kuu + jitter I (posteriors.py:835, covariances/kuus.py:29-34), lower tiles only
  rc = gpk_kernel_matrix(stream, family, Z, m, ldz, nullptr, 0, 0, d, ls_host, ard, variance, jitter,
                         1, T, l.ld);
  if (rc) return rc;
  // Lm = chol(Kuu);  A^T = Kfu Lm^-T   (conditionals/util.py:67,125)
  static const bool out_of_place = getenv("GPK_TRSM_GROUP") != nullptr;
  // streamed projection (see ProjStream): the q_sqrt projection rides along with the extra-row solve, group by group
  const char* sp_env = getenv("GPK_STREAM_PROJ");
  const bool stream_proj = (sp_env ? atoi(sp_env) != 0 : GPK_STREAM_PROJ_DEFAULT) && side && !q_diag && !out_of_place &&
                           (m % GPK_NB) == 0 && getenv("GPK_CHAIN") == nullptr;
  if (out_of_place) {
    rc = gpk_potrf_ex(stream, T, m, rows, l.ld, 1, 0, invd, 0, info, At, l.ld, nullptr, 0);
  } else {
    At = Kfu;  // in place: the extra rows of the trapezoid come back as A^T
    if (stream_proj) {
      g_proj = ProjStream{};
      g_proj.on = true;
      g_proj.LqT = LqT; g_proj.ldl = l.ld; g_proj.strideL = (long)m * l.ld;
      g_proj.C = (double*)(w + l.off_C); g_proj.ldc = l.ld; g_proj.strideC = (long)rows * l.ld;
      g_proj.part = (double*)(w + l.off_proj); g_proj.part_ld = rows; g_proj.stridePart = (long)l.nt * rows;
      g_proj.P = P;
    }
    rc = gpk_potrf(stream, T, m, rows, l.ld, 1, 0, invd, 0, info);
    g_proj.on = false;
  }
  if (rc) return rc;
  const bool projected = stream_proj && g_proj.groups > 0;
  g_proj.groups = 0;
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
