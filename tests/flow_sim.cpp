// CPU simulation of the tile-dataflow schedule (gpflow_amd/csrc/flow_tasks.h): random workgroup scheduling with
// stealing, chain flags raised in order at random moments.  Checks: (1) no deadlock -- every task completes;
// (2) when a task STARTS, its true data dependencies (written out below from the algorithm, not from the list's own
// `need` counters) have completed; (3) read-modify-write tiles are touched by one task at a time and in group order.
#include "flow_tasks.h"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <tuple>

struct Running { int list, idx; int remaining; };

static int simulate(int n, int rows, int P, bool proj, unsigned seed, int workers) {
  std::vector<FlowGroup> groups = flow_groups(n);
  std::vector<FlowTask> lists[8];
  flow_build(n, rows, P, proj, groups, lists);
  const int nrb = (rows + 127) / 128, ng = (int)groups.size();
  std::mt19937 rng(seed);
  std::vector<int> next(8, 0), prog(nrb, 0);
  std::vector<char> flag(ng, 0);
  int flags_up = 0;
  // completion bookkeeping of the TRUE dependency structure
  std::set<std::tuple<int, int, int>> solved;             // (group, rb, tn) solved
  std::map<std::tuple<int, int>, int> e_updates;           // (rb, col tile) -> number of groups whose update was applied
  std::map<std::tuple<int, int, int>, int> c_updates;      // (bz, rb, col tile) -> contributions applied
  std::set<std::tuple<int, int, int>> busy;                // output tiles being written right now (matrix id, rb, tile)
  size_t total = 0, finished = 0;
  for (int x = 0; x < 8; ++x) total += lists[x].size();
  struct W { bool active; int list, idx; bool started; int remaining; };
  std::vector<W> w(workers, W{false, 0, 0, false, 0});
  auto out_tile = [&](const FlowTask& t) {
    const FlowGroup& g = groups[t.group];
    switch (t.type) {
      case FLOW_SOLVE: return std::make_tuple(0, (int)t.rb, g.g0 / 128 + t.tn);
      case FLOW_UPDATE: return std::make_tuple(1, (int)t.rb, g.g1 / 128 + t.tn);
      case FLOW_PROJ_RECT: return std::make_tuple(2 + t.bz, (int)t.rb, (int)t.tn);
      default: return std::make_tuple(2 + t.bz, (int)t.rb, g.g0 / 128 + t.tn);
    }
  };
  auto true_deps_ok = [&](const FlowTask& t) {
    const FlowGroup& g = groups[t.group];
    const int ns = (g.g1 - g.g0) / 128;
    if (t.type == FLOW_SOLVE) {
      if (!flag[t.group]) return false;
      // reads E[rb, column tiles g0/128 .. g0/128 + tn] (triangular K): every earlier group's update on them
      for (int c = g.g0 / 128; c <= g.g0 / 128 + (g.ginv ? t.tn : 0); ++c)
        if (e_updates[{t.rb, c}] != t.group) return false;
      return true;
    }
    for (int tn = 0; tn < ns; ++tn)
      if (!solved.count({t.group, t.rb, tn})) return false;   // A operand: the whole solved group of this row block
    if (t.type == FLOW_UPDATE) {
      if (!flag[t.group]) return false;                        // B operand: L rows below the group (final with the flag)
      return e_updates[{t.rb, g.g1 / 128 + t.tn}] == t.group;  // read-modify-write in group order
    }
    if (t.type == FLOW_PROJ_RECT) {
      // column tile tn < g0/128 has received its own group's triangle and every group in between
      int first = -1;
      for (int gi = 0; gi < ng; ++gi)
        if (groups[gi].g0 / 128 <= t.tn && t.tn < groups[gi].g1 / 128) first = gi;
      return c_updates[{t.bz, t.rb, t.tn}] == t.group - first;
    }
    return c_updates[{t.bz, t.rb, g.g0 / 128 + t.tn}] == 0;    // the triangle is the first write of that tile
  };
  long steps = 0;
  while (finished < total) {
    if (++steps > 50000000L) { std::printf("DEADLOCK n=%d rows=%d P=%d proj=%d seed=%u\n", n, rows, P, (int)proj, seed); return 1; }
    // the chain raises the next flag now and then
    if (flags_up < ng && rng() % 97 == 0) flag[flags_up++] = 1;
    W& me = w[rng() % workers];
    if (!me.active) {  // draw a ticket: own list first, then steal (any list that still has tickets, in ring order)
      const int x = (int)(rng() % 8);
      for (int k = 0; k < 8; ++k) {
        const int xx = (x + k) & 7;
        if (next[xx] < (int)lists[xx].size()) { me = W{true, xx, next[xx]++, false, 0}; break; }
      }
      continue;
    }
    const FlowTask& t = lists[me.list][me.idx];
    if (!me.started) {
      if (t.flag >= 0 && !flag[t.flag]) continue;   // spin
      if (prog[t.rb] < t.need) continue;            // spin
      if (!true_deps_ok(t)) { std::printf("DEPENDENCY VIOLATION type %d group %d rb %d tn %d\n", t.type, t.group, t.rb, t.tn); return 2; }
      if (!busy.insert(out_tile(t)).second) { std::printf("WRITE-WRITE RACE type %d group %d rb %d tn %d\n", t.type, t.group, t.rb, t.tn); return 3; }
      me.started = true;
      me.remaining = 1 + (int)(rng() % 7);
      continue;
    }
    if (--me.remaining > 0) continue;
    // completion
    busy.erase(out_tile(t));
    const FlowGroup& g = groups[t.group];
    if (t.type == FLOW_SOLVE) solved.insert({t.group, t.rb, t.tn});
    else if (t.type == FLOW_UPDATE) e_updates[{t.rb, g.g1 / 128 + t.tn}]++;
    else if (t.type == FLOW_PROJ_RECT) c_updates[{t.bz, t.rb, t.tn}]++;
    else c_updates[{t.bz, t.rb, g.g0 / 128 + t.tn}]++;
    prog[t.rb]++;
    ++finished;
    me.active = false;
  }
  // every output column tile of the projection received all its contributions
  if (proj)
    for (int bz = 0; bz < P; ++bz)
      for (int rb = 0; rb < nrb; ++rb)
        for (int c = 0; c < n / 128; ++c) {
          int first = -1;
          for (int gi = 0; gi < ng; ++gi)
            if (groups[gi].g0 / 128 <= c && c < groups[gi].g1 / 128) first = gi;
          if (c_updates[{bz, rb, c}] != ng - first) { std::printf("PROJECTION INCOMPLETE\n"); return 4; }
        }
  return 0;
}

int main() {
  const int shapes[][4] = {{2048, 8192, 1, 1}, {1024, 8192, 1, 1}, {1024, 2500, 4, 1}, {512, 300, 2, 1}, {1152, 777, 3, 1},
                           {640, 1000, 2, 1}, {2048, 1024, 1, 0}, {1536, 513, 1, 1}};
  for (const auto& s : shapes)
    for (unsigned seed = 1; seed <= 3; ++seed)
      for (int workers : {1, 7, 448}) {
        const int rc = simulate(s[0], s[1], s[2], s[3] != 0, seed, workers);
        if (rc) return rc;
      }
  std::vector<FlowTask> lists[8];
  flow_build(2048, 8192, 1, true, flow_groups(2048), lists);
  size_t total = 0;
  for (auto& l : lists) total += l.size();
  std::printf("flow schedule ok (Cm: %zu groups, %zu tasks)\n", flow_groups(2048).size(), total);
  return 0;
}
