// Task list of the tile-dataflow bulk kernel (gemm.hip: flow_kernel) -- host-only, header-only, no HIP types, so that
// the CPU test tier can compile it with g++ and SIMULATE the schedule (tests/flow_sim.cpp: no deadlock, every tile
// starts after its true producers, read-modify-write tiles in order).
//
// What it describes.  Beside the latency chain of an SVGP-sized factorisation (leaf -> panel solve -> strip on the
// reserved compute units) ALL the bulk work of the step -- the right-looking solve of the minibatch rows and the
// streamed projection onto q_sqrt -- is ONE persistent launch whose workgroups draw 128 x 128 output tiles from per-XCD
// ticket lists and wait, per tile, for exactly what that tile needs:
//   * a chain flag: the columns of group g of the factor (and the group's explicit inverse) exist;
//   * a row-block counter: the earlier stages of the SAME 128-row block have finished (rows are independent, so the
//     whole dependency structure is 64 independent pipelines gated by the chain).
// Per row block rb and column group g = [g0, g1) the stages are
//   A_g  solve    S[rb, g]   = E[rb, g] * inv(L[g, g])^T           (w / 128 tiles; K <= w, triangular)
//   B_g  update   E[rb, c]  -= S[rb, g] * L[c, g]^T   for c > g    ((n - g1) / 128 tiles; K = w)
//        project  C[rb, i] (+)= S[rb, g] * LqT[i, g]^T for i <= g  (P * g1 / 128 tiles; K = w, triangular on the diagonal)
// and a task of stage k of rb needs prog[rb] >= (number of tasks of rb in stages < k).  Lists are stage-major, so every
// dependency of a ticket is an EARLIER ticket of the same list or a chain flag: workgroups that take tickets in order
// can always make progress as long as the chain does.
#pragma once
#include <stdint.h>
#include <vector>

#define GPK_FLOW_MAX_GROUPS 24

struct FlowGroup {
  int g0, g1;
  int ginv;  // 1: up to 512 columns solved against the group's explicit inverse; 0: 128 columns, the leaf's block inverse
};

enum { FLOW_SOLVE = 0, FLOW_UPDATE = 1, FLOW_PROJ_RECT = 2, FLOW_PROJ_TRI = 3 };

struct FlowTask {  // 16 bytes
  uint8_t type;
  uint8_t group;
  uint8_t bz;    // latent GP (projection tasks)
  uint8_t last;  // projection task of the LAST group: squares and row-sums instead of storing
  uint16_t rb;   // row block of 128 rows
  uint16_t tn;   // column tile inside the task's GEMM
  int32_t need;  // prog[rb] must have reached this value
  int32_t flag;  // index of the chain flag that must carry the current epoch, or -1
};

// column groups of an n-column factor (n % 128 == 0, n >= 512):
//   * a first group of 256 columns: the bulk kernel can start after two panels of the chain instead of four;
//   * 512-column groups (K = 512 tiles are the efficient ones) up to column n - 512;
//   * the last 512 columns as two groups of 256: whatever bulk work is left when the LAST leaf finishes is exposed
//     latency -- but single 128-column blocks there meant 3456 projection tiles with K = 128 (a quarter of the step's
//     flops at a third of the kernel's rate: 0.9 ms after the chain had ended, measured).
// Groups wider than one block (ginv = 1, width <= 512) are solved against their explicit inverse, single blocks against
// the leaf's block inverse.
inline std::vector<FlowGroup> flow_groups(int n) {
  std::vector<FlowGroup> g;
  const int tail = n - 512 > 0 ? n - 512 : 0;  // columns [tail, n): two groups of 256
  int c = 0;
  if (tail >= 256) {
    g.push_back({0, 256, 1});
    c = 256;
  }
  while (c < tail) {
    const int w = (tail - c >= 512) ? 512 : tail - c;
    g.push_back({c, c + w, w > 128 ? 1 : 0});
    c += w;
  }
  while (c < n) {
    const int w = (n - c >= 256) ? 256 : n - c;
    g.push_back({c, c + w, w > 128 ? 1 : 0});
    c += w;
  }
  return g;
}

// per-XCD ticket lists (row block rb belongs to list rb % 8: its A operand then stays in that XCD's L2)
inline void flow_build(int n, int rows, int P, bool proj, const std::vector<FlowGroup>& groups,
                       std::vector<FlowTask> lists[8]) {
  const int nrb = (rows + 127) / 128;
  std::vector<int> done(nrb, 0);  // tasks of rb in completed stages
  for (int x = 0; x < 8; ++x) lists[x].clear();
  const int ng = (int)groups.size();
  for (int gi = 0; gi < ng; ++gi) {
    const FlowGroup& g = groups[gi];
    const int w = g.g1 - g.g0, ns = (w + 127) / 128, nu = (n - g.g1 + 127) / 128, nr = g.g0 / 128;
    const bool last = gi == ng - 1;
    // stage A_g
    for (int rb = 0; rb < nrb; ++rb)
      for (int t = 0; t < ns; ++t)
        lists[rb & 7].push_back({FLOW_SOLVE, (uint8_t)gi, 0, 0, (uint16_t)rb, (uint16_t)t, done[rb], gi});
    for (int rb = 0; rb < nrb; ++rb) done[rb] += ns;
    // stage B_g
    for (int rb = 0; rb < nrb; ++rb) {
      for (int t = 0; t < nu; ++t)
        lists[rb & 7].push_back({FLOW_UPDATE, (uint8_t)gi, 0, 0, (uint16_t)rb, (uint16_t)t, done[rb], -1});
      if (proj)
        for (int bz = 0; bz < P; ++bz) {
          for (int t = 0; t < nr; ++t)
            lists[rb & 7].push_back({FLOW_PROJ_RECT, (uint8_t)gi, (uint8_t)bz, (uint8_t)last, (uint16_t)rb, (uint16_t)t,
                                     done[rb], -1});
          for (int t = 0; t < ns; ++t)
            lists[rb & 7].push_back({FLOW_PROJ_TRI, (uint8_t)gi, (uint8_t)bz, (uint8_t)last, (uint16_t)rb, (uint16_t)t,
                                     done[rb], -1});
        }
    }
    for (int rb = 0; rb < nrb; ++rb) done[rb] += nu + (proj ? P * (nr + ns) : 0);
  }
}
