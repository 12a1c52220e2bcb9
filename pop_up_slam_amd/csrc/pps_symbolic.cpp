// pps_symbolic.cpp -- see pps_symbolic.h.
#include "pps_symbolic.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdio>
#include <functional>
#include <numeric>
#include <queue>

namespace pps {
namespace {

struct TNode {
  std::vector<int> piv;   // node ids
  std::vector<int> kids;  // tnode ids
};

// One dissected sub-chain of a previous analysis: the poses [first .. last] (ids; a chain only grows at its end, so first /
// last / count identify the set), the planes handed down to it, and the block of tree nodes [t0, t1) it produced.
struct DissectMemo { int first, last, count, t0, t1; std::vector<int> planes; };

struct Builder {
  const std::vector<SymNode>& nodes;
  const std::vector<SymFactor>& factors;
  const AnalysisParams& prm;
  int N;
  // frame-loop reuse (aligned cuts only): sub-chains of the previous analysis by (first pose, last pose); a sub-chain that
  // comes back with the same poses and planes is the same sub-tree -- nothing inside it has a new neighbour -- and its
  // block of tree nodes is copied instead of dissected again
  std::vector<TNode>* old_tree = nullptr;        // (not const: reused tree nodes and memos are MOVED out, the cache is rebuilt at the end)
  std::vector<DissectMemo>* old_memo = nullptr;
  const std::vector<int>* old_memo_of_first = nullptr;   // pose id -> first memo index of the sub-chains starting there (-1)
  const std::vector<int>* old_memo_next = nullptr;       // memo index -> next memo with the same first pose (-1)
  const std::vector<char>* touched = nullptr;            // old nodes a new factor is attached to: their sub-trees' boundaries gain a node
  std::vector<DissectMemo> memo;                         // of THIS run
  std::vector<char> t_reused;                            // tree node copied from the previous analysis
  std::vector<int> t_old;                                // ... and which one it was there
  std::vector<int> adj_off, adj;   // CSR, unique neighbours
  std::vector<TNode> tree;
  std::vector<int> lidx;           // scratch: node -> local pose index in the current call
  std::vector<char> dense;
  int chain_first = -1, chain_last = -1;   // first / last pose of the whole chain (node ids)
  int border_dim = 0;              // scalars of the dense border: rows of every front (the ground plane is seen from every pose)
  std::vector<int> own_stamp;      // scratch of ext_adjacent
  int own_stamp_id = 0;

  Builder(const std::vector<SymNode>& n, const std::vector<SymFactor>& f, const AnalysisParams& p)
      : nodes(n), factors(f), prm(p), N((int)n.size()) {}

  void build_adjacency() {
    // CSR of the undirected node graph, neighbours sorted and unique: counting sort on the first end point, small
    // sorts inside each row
    static thread_local std::vector<int> cnt, raw, fill;    // scratch that keeps its capacity from call to call (frame loops)
    cnt.assign(N + 1, 0);
    for (const auto& f : factors)
      if (f.b >= 0 && f.a != f.b) { cnt[f.a + 1]++; cnt[f.b + 1]++; }
    for (int i = 0; i < N; i++) cnt[i + 1] += cnt[i];
    raw.resize(cnt[N]); fill.assign(cnt.begin(), cnt.end() - 1);
    for (const auto& f : factors)
      if (f.b >= 0 && f.a != f.b) { raw[fill[f.a]++] = f.b; raw[fill[f.b]++] = f.a; }
    adj_off.assign(N + 1, 0);
    adj.clear();
    adj.reserve(raw.size());
    for (int u = 0; u < N; u++) {
      int* b = raw.data() + cnt[u];
      int* e = raw.data() + cnt[u + 1];
      if (!std::is_sorted(b, e)) std::sort(b, e);      // factors arrive in time order: a landmark's observers already are
      e = std::unique(b, e);
      adj.insert(adj.end(), b, e);
      adj_off[u + 1] = (int)adj.size();
    }
  }
  int degree(int u) const { return adj_off[u + 1] - adj_off[u]; }

  // The same adjacency from the one of the previous analysis (rows of the first n0 nodes, edges of the first m0 factors) and the
  // factors behind m0, every one of which touches a node >= n0 (checked by the caller): an old row gains neighbours >= n0 only, i.e.
  // behind everything it holds -- one copy of the old rows with the sorted additions appended, instead of a counting sort over
  // every factor of the graph on every frame.  keep(u, v): rows / entries to leave out (the dense border for the dissection's view).
  template <class Keep>
  void extend_adjacency(const std::vector<int>& off0, const std::vector<int>& adj0, size_t n0, size_t m0, std::vector<int>& off1,
                        std::vector<int>& adj1, Keep keep) const {
    static thread_local std::vector<int> cnt, raw, fill;
    cnt.assign(N + 1, 0);
    for (size_t i = m0; i < factors.size(); i++) {
      const auto& f = factors[i];
      if (f.b >= 0 && f.a != f.b) { cnt[f.a + 1]++; cnt[f.b + 1]++; }
    }
    for (int i = 0; i < N; i++) cnt[i + 1] += cnt[i];
    raw.resize(cnt[N]); fill.assign(cnt.begin(), cnt.end() - 1);
    for (size_t i = m0; i < factors.size(); i++) {
      const auto& f = factors[i];
      if (f.b >= 0 && f.a != f.b) { raw[fill[f.a]++] = f.b; raw[fill[f.b]++] = f.a; }
    }
    off1.assign(N + 1, 0);
    adj1.clear();
    adj1.reserve(adj0.size() + raw.size());
    for (int u = 0; u < N; u++) {
      if ((size_t)u < n0) adj1.insert(adj1.end(), adj0.begin() + off0[u], adj0.begin() + off0[u + 1]);
      int* b = raw.data() + cnt[u];
      int* e = raw.data() + cnt[u + 1];
      if (b != e) {
        if (!std::is_sorted(b, e)) std::sort(b, e);
        e = std::unique(b, e);
        for (int* q = b; q < e; q++) if (keep(u, *q)) adj1.push_back(*q);
      }
      off1[u + 1] = (int)adj1.size();
    }
  }
  // any pose-pose edge between poses that are not neighbours in rank?  (none in a frame loop: dissect() then has no cross edges to
  // look for in any sub-chain -- two poses of consecutive rank are neighbours in every sub-chain that holds both)
  bool cross_edges_among(size_t m0) const {
    for (size_t i = m0; i < factors.size(); i++) {
      const auto& f = factors[i];
      if (f.b < 0 || f.a == f.b || nodes[f.a].type != NODE_POSE || nodes[f.b].type != NODE_POSE) continue;
      const int dr = nodes[f.a].rank - nodes[f.b].rank;
      if (dr != 1 && dr != -1) return true;
    }
    return false;
  }
  bool any_cross = true;           // (false: proven that no sub-chain has a pose-pose edge between non-neighbours)

  int new_tnode() { tree.emplace_back(); t_reused.push_back(0); t_old.push_back(-1); return (int)tree.size() - 1; }

  // General fill-reducing ordering for graphs the pose chain does not dissect well (pose graphs with many loop
  // closures, 2-D meshes): minimum degree on the node graph (degrees in scalars, lazy heap, adjacency lists merged
  // at every elimination), elimination tree from the filled column structures, fundamental supernodes (a node joins
  // its parent when it is the parent's only child and their structures coincide).  `live` = the nodes to order (the
  // current adjacency must already be restricted to them).  Returns the root tnode id (several components: the last
  // root adopts the others).
  int mindeg_tree(const std::vector<int>& live) {
    std::vector<std::vector<int>> nb(N);
    for (int u : live) nb[u].assign(adj.begin() + adj_off[u], adj.begin() + adj_off[u + 1]);
    std::vector<int> deg(N, 0);
    std::vector<char> gone(N, 1);
    using HE = std::pair<int, int>;
    std::priority_queue<HE, std::vector<HE>, std::greater<HE>> heap;
    for (int u : live) {
      gone[u] = 0;
      int d = 0;
      for (int v : nb[u]) d += nodes[v].dim;
      deg[u] = d;
      heap.emplace(d, u);
    }
    std::vector<int> order, pos(N, -1);
    std::vector<std::vector<int>> strct(N);     // filled structure of the column at elimination time
    std::vector<int> tmp;
    while (!heap.empty()) {
      const HE e = heap.top(); heap.pop();
      const int v = e.second;
      if (gone[v] || e.first != deg[v]) continue;    // stale entry
      gone[v] = 1;
      pos[v] = (int)order.size();
      order.push_back(v);
      strct[v].swap(nb[v]);
      const std::vector<int>& Nv = strct[v];
      for (int u : Nv) {
        std::vector<int>& U = nb[u];
        tmp.clear();
        size_t i = 0, j = 0;
        while (i < U.size() || j < Nv.size()) {        // sorted merge, dropping u and v
          int x;
          if (j >= Nv.size() || (i < U.size() && U[i] <= Nv[j])) { x = U[i]; if (j < Nv.size() && Nv[j] == x) j++; i++; }
          else { x = Nv[j]; j++; }
          if (x != u && x != v) tmp.push_back(x);
        }
        U.assign(tmp.begin(), tmp.end());
        int d = 0;
        for (int x : U) d += nodes[x].dim;
        deg[u] = d;
        heap.emplace(d, u);
      }
    }
    const int n = (int)order.size();
    // elimination tree: the parent of v is the member of its structure that is eliminated first
    std::vector<int> parent(N, -1), nchild(N, 0);
    for (int v : order) {
      int best = -1;
      for (int u : strct[v]) if (best < 0 || pos[u] < pos[best]) best = u;
      parent[v] = best;
      if (best >= 0) nchild[best]++;
    }
    // supernodes along the elimination order
    // (relaxed: an only child also joins when the parent's structure is at most a quarter larger -- a few explicit
    // zeros in the child's columns buy fewer, fuller fronts and a shallower tree -- up to max_pivots scalars per front)
    std::vector<int> sn_of(N, -1), sn_dim;
    for (int k = 0; k < n; k++) {
      const int v = order[k];
      if (sn_of[v] < 0) { sn_of[v] = new_tnode(); sn_dim.resize(tree.size(), 0); }
      tree[sn_of[v]].piv.push_back(v);
      sn_dim[sn_of[v]] += nodes[v].dim;
      const int par = parent[v];
      if (par >= 0 && nchild[par] == 1 && pos[par] == pos[v] + 1) {
        const size_t sv = strct[v].size(), sp = strct[par].size() + 1;      // struct(v) is a subset of {par} + struct(par)
        const bool fundamental = sp == sv;
        const bool relaxed = sp <= sv + std::max<size_t>(2, sv / 4) && sn_dim[sn_of[v]] + nodes[par].dim <= prm.max_pivots;
        if (fundamental || relaxed) sn_of[par] = sn_of[v];
      }
    }
    // supernode tree
    std::vector<int> roots;
    for (int k = 0; k < n; k++) {
      const int v = order[k];
      const int t = sn_of[v];
      if (tree[t].piv.back() != v) continue;           // only the last node of a supernode links upwards
      const int par = parent[v];
      if (par < 0) roots.push_back(t);
      else tree[sn_of[par]].kids.push_back(t);
    }
    if (roots.empty()) return -1;
    const int top = roots.back();
    for (size_t r = 0; r + 1 < roots.size(); r++) tree[top].kids.push_back(roots[r]);
    return top;
  }

  // The nodes of `cand` (separator nodes of the ancestors) that have a neighbour among `poses` / `planes`: what the fronts of this
  // sub-chain's sub-tree carry as boundary rows besides the border.  A function of the sub-chain and the adjacency alone (its external
  // neighbours are exactly these), so a memoised sub-tree needs no record of it.
  std::vector<int> ext_adjacent(const std::vector<int>& cand, const std::vector<int>& poses, const std::vector<int>& planes, int* dim) {
    std::vector<int> out;
    *dim = 0;
    if (cand.empty() || poses.empty()) return out;
    if ((int)own_stamp.size() < N) own_stamp.assign(N, 0);
    const int id = ++own_stamp_id;
    int lo = poses[0], hi = poses[0];
    for (int u : poses) { own_stamp[u] = id; lo = std::min(lo, u); hi = std::max(hi, u); }
    for (int u : planes) { own_stamp[u] = id; lo = std::min(lo, u); hi = std::max(hi, u); }
    for (int v : cand) {
      const int* a0 = adj.data() + adj_off[v];
      const int* a1 = adj.data() + adj_off[v + 1];
      bool hit = false;
      for (const int* q = std::lower_bound(a0, a1, lo); q < a1 && *q <= hi && !hit; q++) hit = own_stamp[*q] == id;      // (rows are sorted)
      if (hit) { out.push_back(v); *dim += nodes[v].dim; }
    }
    return out;
  }

  // poses: node ids sorted by rank; planes: node ids; ext: separator nodes of the ancestors (candidates for this sub-chain's external
  // boundary).  Returns tnode id.
  int dissect(std::vector<int> poses, std::vector<int> planes, const std::vector<int>& ext = std::vector<int>()) {
    const int n = (int)poses.size();
    if (old_memo && n > 0 && poses[0] < (int)old_memo_of_first->size()) {
      for (int mi = (*old_memo_of_first)[poses[0]]; mi >= 0; mi = (*old_memo_next)[mi]) {
        const DissectMemo& m = (*old_memo)[mi];
        if (m.last != poses[n - 1] || m.count != n || m.planes != planes) continue;
        // ... none of whose nodes has a new neighbour: a new factor may hang on an old node that stays where it was -- the last old
        // pose when the first new one becomes a cut (several frames between two analyses), a pose a new one closes a loop with, a
        // plane seen again from a cut pose -- and the fronts of that node's sub-tree then gain a boundary node
        if (touched) {
          bool hit = false;
          for (int u : poses) hit = hit || (*touched)[u];
          for (int u : planes) hit = hit || (*touched)[u];
          if (hit) continue;
        }
        // the same sub-chain with the same planes: copy its block of tree nodes
        const int t0 = (int)tree.size(), delta = t0 - m.t0, m_t0 = m.t0, m_t1 = m.t1;
        for (int q = m_t0; q < m_t1; q++) {
          tree.push_back(std::move((*old_tree)[q]));
          for (int& k : tree.back().kids) k += delta;
          t_reused.push_back(1); t_old.push_back(q);
        }
        // the memos of the copied block stay valid for the next frame (re-based): memos are pushed in pre-order, so the
        // block's own are the run that starts at this one
        for (size_t k = (size_t)mi; k < old_memo->size(); k++) {
          DissectMemo& mm = (*old_memo)[k];
          if (mm.t0 < m_t0 || mm.t1 > m_t1) break;
          memo.push_back(std::move(mm)); memo.back().t0 += delta; memo.back().t1 += delta;
          mm.t0 = mm.t1 = -1; mm.count = -1;           // moved from: never matches again
        }
        return t0;
      }
    }
    const int t = new_tnode();
    const size_t my_memo = memo.size();
    memo.push_back(DissectMemo{n > 0 ? poses[0] : -1, n > 0 ? poses[n - 1] : -1, n, t, -1, planes});
    // rows the first front of this sub-chain's top would hold besides its own pivots
    int ext_dim = 0;
    std::vector<int> ext_here;
    const bool limit = prm.front_rows > 0;
    if (limit) ext_here = ext_adjacent(ext, poses, planes, &ext_dim);
    const int fixed_rows = ext_dim + border_dim;
    int own_dim = 0;
    for (int po : poses) own_dim += nodes[po].dim;
    for (int pl : planes) own_dim += nodes[pl].dim;
    // (a would-be leaf of more than front_rows rows is dissected further: its poses leave it one separator at a time)
    if (n <= prm.leaf_poses && (!limit || n <= 1 || own_dim + fixed_rows <= prm.front_rows)) {
      for (int pl : planes) tree[t].piv.push_back(pl);
      for (int po : poses) tree[t].piv.push_back(po);
      memo[my_memo].t1 = (int)tree.size();
      return t;
    }
    for (int i = 0; i < n; i++) lidx[poses[i]] = i;
    // observer span of every plane inside this sub-chain
    std::vector<int> pmin(planes.size(), n), pmax(planes.size(), -1);
    std::vector<int> diff(n + 1, 0);
    std::vector<int> only(limit ? n : 0, 0);               // scalars of the planes seen from pose m alone (they join a cut at m)
    // A landmark's observers are sorted by node id, which grows with the pose rank: inside this sub-chain they are one run of
    // the row, and its two ends are the span (two binary searches instead of a walk over every observer -- the ground-level
    // chains of a frame loop are walked once per frame).  Anything unexpected falls back to the walk.
    const bool ids_sorted = std::is_sorted(poses.begin(), poses.end());
    for (size_t k = 0; k < planes.size(); k++) {
      const int pl = planes[k];
      const int* a0 = adj.data() + adj_off[pl];
      const int* a1 = adj.data() + adj_off[pl + 1];
      bool done = false;
      if (ids_sorted) {
        const int* f = std::lower_bound(a0, a1, poses[0]);
        const int* l = std::upper_bound(f, a1, poses[n - 1]);
        if (f == l) done = true;                                  // not seen from this sub-chain
        else if (lidx[*f] >= 0 && lidx[*(l - 1)] >= 0) { pmin[k] = lidx[*f]; pmax[k] = lidx[*(l - 1)]; done = true; }
      }
      if (!done)
        for (const int* q = a0; q < a1; q++) {
          const int li = lidx[*q];
          if (li < 0) continue;
          pmin[k] = std::min(pmin[k], li);
          pmax[k] = std::max(pmax[k], li);
        }
      if (pmax[k] - pmin[k] >= 2) { diff[pmin[k] + 1] += 3; diff[pmax[k]] -= 3; }  // spans m for pmin < m < pmax
      if (limit && pmax[k] == pmin[k] && pmin[k] >= 0 && pmin[k] < n) only[pmin[k]] += nodes[pl].dim;
    }
    // pose-pose edges that are not between chain neighbours
    std::vector<std::pair<int, int>> cross;
    for (int i = 0; any_cross && i < n; i++) {
      const int u = poses[i];
      for (int q = adj_off[u]; q < adj_off[u + 1]; q++) {
        const int v = adj[q];
        if (nodes[v].type != NODE_POSE) continue;
        const int lj = lidx[v];
        if (lj > i + 1) { cross.emplace_back(i, lj); diff[i + 1] += 6; diff[lj] -= 6; }
      }
    }
    std::vector<int> cost(n, 0);
    { int run = 0; for (int m2 = 0; m2 < n; m2++) { run += diff[m2]; cost[m2] = run; } }
    // cut positions: arity-1 cuts near the quantiles, each moved inside its window to the cheapest position
    int K = std::max(2, prm.arity);
    while (K > 2 && n < 2 * K + 1) K--;        // too short for that many parts
    std::vector<int> cuts;
    const bool aligned = K == 2 && prm.aligned_cuts;
    if (aligned) {
      // Bisection at an ABSOLUTE position: the pose whose rank (insertion index) is the multiple of the largest power of two
      // inside this sub-chain.  A chain that grows at its end (a SLAM front
      // end adds one pose per frame) then keeps every sub-tree left of its newest poses -- ordering, fronts and all index
      // arrays of that part stay what they were, frame after frame (a quantile cut moves with n).
      // (round 6: cuts may leave their aligned rank by a few poses -- below -- so the ends of a sub-chain are no longer aligned themselves, and
      // the multiple of the largest power of two inside it can sit next to one of them.  The rank is therefore looked for in the MIDDLE QUARTER
      // of the sub-chain -- still an absolute position, a function of the sub-chain's two end ranks alone -- measured from the ends that are
      // cuts: the chain's own first pose bounds nothing, and the newest poses of a growing chain must not push the cut of the right spine back)
      int r_lo = nodes[poses[0]].rank, r_hi = nodes[poses[n - 1]].rank;
      if (limit && prm.cut_shift > 0) {
        const int margin = 3 * (r_hi - r_lo) / 8;
        if (poses[0] != chain_first) r_lo += margin;
        if (poses[n - 1] != chain_last) r_hi -= margin;
      }
      int c_rank = r_hi;
      for (int k = 30; k >= 0; k--) { const int c = (r_hi >> k) << k; if (c > r_lo) { c_rank = c; break; } }
      int centre = (int)(std::lower_bound(poses.begin(), poses.end(), c_rank, [&](int u, int r) { return nodes[u].rank < r; }) - poses.begin());
      // (no search window here: a cut that leaves its aligned rank makes the child ranges straddle their own aligned cuts
      // and the tree degenerates -- 12 levels instead of 9 on a 512-pose chain)
      centre = std::min(std::max(centre, 1), n - 2);
      // (round 6) ... except by a few poses: within cut_shift of the aligned rank the cheapest position is taken (ties: the nearest).  The
      // planes of a cut are boundary rows of every front below it on that side, so a sub-chain between two cuts of five walls each whose own
      // cut holds five more is a front of 66 rows; a pose that sees four walls is rarely far.  The aligned rank itself keeps away from the
      // ends of the sub-chain (above), so the children never find their parent's rank inside their own window.
      if (limit && prm.cut_shift > 0 && cross.empty()) {
        int best = centre;
        for (int dlt = 1; dlt <= std::min(prm.cut_shift, n / 8); dlt++)
          for (int sg = -1; sg <= 1; sg += 2) {
            const int m2 = centre + sg * dlt;
            if (m2 < 1 || m2 > n - 2) continue;
            if (cost[m2] + only[m2] < cost[best] + only[best]) best = m2;
          }
        centre = best;
      }
      cuts.push_back(centre);
    }
    for (int c = 1; c < K && !aligned; c++) {
      const int centre = (int)((long long)n * c / K);
      const int half = std::max(0, n / (6 * K));
      int lo = std::max(1, centre - half), hi = std::min(n - 2, centre + half);
      if (!cuts.empty()) lo = std::max(lo, cuts.back() + 2);
      if (hi < lo) { if (lo <= n - 2) hi = lo; else continue; }
      int best = lo, best_cost = cost[lo];
      for (int m2 = lo; m2 <= hi; m2++)
        if (cost[m2] < best_cost || (cost[m2] == best_cost && std::abs(m2 - centre) < std::abs(best - centre))) { best_cost = cost[m2]; best = m2; }
      cuts.push_back(best);
    }
    if (cuts.empty()) cuts.push_back(std::min(std::max(1, n / 2), n - 1));
    if (limit && cuts.size() == 1 && cross.empty() && n >= 3) {
      // The separator front of a cut at m holds the cut pose, the planes that span m or are seen from m alone, and the sub-chain's external
      // boundary.  Beyond front_rows the cut moves to the NEAREST position that fits, at most a quarter of the sub-chain away (the parts stay
      // balanced; an aligned cut has already taken the cheapest position within cut_shift poses of its rank, so this is its last resort); where
      // nothing in reach fits it stays.
      auto rows_at = [&](int m2) { return nodes[poses[m2]].dim + cost[m2] + only[m2] + fixed_rows; };
      const int m0 = cuts[0];
      if (rows_at(m0) > prm.front_rows) {
        const int reach = std::max(1, n / 4);
        bool fits = false;
        for (int dlt = 1; dlt <= reach && !fits; dlt++)
          for (int sg = -1; sg <= 1 && !fits; sg += 2) {
            const int m2 = m0 + sg * dlt;
            if (m2 < 1 || m2 > n - 2) continue;
            if (rows_at(m2) <= prm.front_rows) { cuts[0] = m2; fits = true; }
          }
      }
    }
    const int nparts = (int)cuts.size() + 1;
    // part index of every pose (-1 = separator)
    std::vector<int> part(n, 0);
    {
      size_t ci = 0;
      for (int i = 0; i < n; i++) {
        while (ci < cuts.size() && i > cuts[ci]) ci++;
        part[i] = (ci < cuts.size() && i == cuts[ci]) ? -1 : (int)ci;
      }
    }
    for (auto& e : cross)
      if (part[e.first] != part[e.second] && part[e.first] >= 0 && part[e.second] >= 0) part[e.second] = -1;
    std::vector<std::vector<int>> pposes(nparts), pplanes(nparts);
    std::vector<int> sep_planes, sep_poses;
    for (int i = 0; i < n; i++) {
      if (part[i] < 0) sep_poses.push_back(poses[i]);
      else pposes[part[i]].push_back(poses[i]);
    }
    const bool one_cut = cuts.size() == 1 && cross.empty();     // parts = [0, cut) and (cut, n): the spans above decide
    for (size_t k = 0; k < planes.size(); k++) {
      const int pl = planes[k];
      if (one_cut) {
        const bool left = pmin[k] < cuts[0], right = pmax[k] > cuts[0];
        if (left == right) sep_planes.push_back(pl);             // spans the cut, or attached to the cut pose (or ancestors) only
        else pplanes[left ? 0 : 1].push_back(pl);
        continue;
      }
      // which parts still hold an observer once the separator poses are gone?
      int seen = -1; bool multi = false;
      for (int q = adj_off[pl]; q < adj_off[pl + 1]; q++) {
        const int li = lidx[adj[q]];
        if (li < 0 || part[li] < 0) continue;
        if (seen < 0) seen = part[li];
        else if (seen != part[li]) { multi = true; break; }
      }
      if (multi || seen < 0) sep_planes.push_back(pl);   // spans a cut, or attached to separator poses (or ancestors) only
      else pplanes[seen].push_back(pl);
    }
    for (int i = 0; i < n; i++) lidx[poses[i]] = -1;
    for (int pl : sep_planes) tree[t].piv.push_back(pl);
    for (int po : sep_poses) tree[t].piv.push_back(po);
    std::vector<int> ext_kids;
    if (limit) { ext_kids = ext_here; ext_kids.insert(ext_kids.end(), sep_planes.begin(), sep_planes.end()); ext_kids.insert(ext_kids.end(), sep_poses.begin(), sep_poses.end()); }
    for (int q = 0; q < nparts; q++)
      if (!pposes[q].empty()) { const int c = dissect(std::move(pposes[q]), std::move(pplanes[q]), ext_kids); tree[t].kids.push_back(c); }
    memo[my_memo].t1 = (int)tree.size();
    return t;
  }
};

}  // namespace

// A fresh analysis in an Analysis that has been used before: every vector keeps its capacity, so a frame loop (one new pose
// per call, every array a little longer than last time) does not go back to the allocator for megabytes per frame.
void reset_keep_capacity(Analysis& A) {
  A.kept = Analysis::Kept();
  A.node_pos.clear();
  A.node_voff.clear();
  A.order.clear();
  A.f_p.clear();
  A.f_b.clear();
  A.f_poff.clear();
  A.f_parent.clear();
  A.f_level.clear();
  A.f_Loff.clear();
  A.f_Uoff.clear();
  A.f_bidx_off.clear();
  A.bidx.clear();
  A.f_child_off.clear();
  A.child.clear();
  A.f_cmap_off.clear();
  A.cmap.clear();
  A.level_off.clear();
  A.level_fronts.clear();
  A.f_asm_off.clear();
  A.asm_blk.clear();
  A.asm_lrow.clear();
  A.asm_lcol.clear();
  A.asm_el0.clear();
  A.asm_fsz.clear();
  A.stage_grp_off.clear();
  A.grp_lvl_off.clear();
  A.glvl_front_off.clear();
  A.glvl_fronts.clear();
  A.stage_max_front.clear();
  A.stage_max_width.clear();
  A.blk_doff.clear();
  A.blk_dst.clear();
  A.f_el_off.clear();
  A.el_src.clear();
  A.el_tgt.clear();
  A.f_ea_off.clear();
  A.ea_tgt.clear();
  A.frec.clear();
  A.crec.clear();
  A.srec.clear();
  A.blk_rows.clear();
  A.blk_cols.clear();
  A.blk_size.clear();
  A.blk_nseg.clear();
  A.blk_hoff.clear();
  A.seg_blk.clear();
  A.seg_c0.clear();
  A.seg_cnt.clear();
  A.seg_hoff.clear();
  A.contrib.clear();
  A.n_nodes = A.n_scalars = 0;
  A.n_fronts = A.n_levels = A.max_front = 0;
  A.el_total = 0; A.L_size = A.U_size = 0;
  A.n_stages = A.n_groups = A.n_glevels = 0;
  A.ea_total = 0;
  A.n_blocks = 0; A.n_segs = 0;
  A.H_size = 0; A.J_size = 0;
  A.obs_dir.clear(); A.nd_segs.clear(); A.pidx.clear();
}

// What an analysis leaves behind for the next analysis of the same, grown graph (see pps_symbolic.h).
struct AnalysisCache {
  bool valid = false;
  std::vector<SymNode> nodes;
  std::vector<SymFactor> factors;
  AnalysisParams prm;
  std::vector<char> dense;
  std::vector<TNode> tree;
  std::vector<DissectMemo> memo;
  std::vector<int> memo_of_first, memo_next;
  std::vector<int> post;                  // post-order sequence of tree nodes
  std::vector<int> post_fend;             // fronts emitted up to and including post[k]
  std::vector<int> f_pos0, f_npiv;        // first position / node count of every front
  std::vector<std::vector<int>> bnd;      // boundary nodes of every front, in elimination order
  std::vector<int> blk_pu;                // column position of every H block (blocks are sorted by it)
  std::vector<int> adj_off, adj;          // node adjacency (CSR, sorted unique) ...
  std::vector<int> adj2_off, adj2;        // ... and the dissection's view of it (dense border left out)
  bool any_cross = true;                  // a pose-pose edge between poses that are not rank neighbours exists
  std::vector<std::vector<int>> inc;      // factors of every node, in factor order, for the first inc_factors factors
  size_t inc_factors = 0;
  int64_t J_size = 0, P_size = 0;         // running maxima over those factors
  int n_obs_slots = 0;
  int fronts_reused = 0, fronts_total = 0;
};
AnalysisCache* analysis_cache_new() { return new AnalysisCache(); }
void analysis_cache_free(AnalysisCache* c) { delete c; }
void analysis_cache_stats(const AnalysisCache* c, int* fronts_reused, int* fronts_total) {
  if (fronts_reused) *fronts_reused = c ? c->fronts_reused : 0;
  if (fronts_total) *fronts_total = c ? c->fronts_total : 0;
}

namespace {
// 1 = done, 0 = error (msg), 2 = the reuse attempt does not apply after all: run again from scratch
int analyze_with(const std::vector<SymNode>& nodes, const std::vector<SymFactor>& factors, const AnalysisParams& prm,
                 Analysis& A, const char** msg, bool general_ordering, AnalysisCache* C, bool reuse) {
  static const char* kOk = "";
  *msg = kOk;
  const bool timing = prm.timing != 0;
  auto t_prev = std::chrono::steady_clock::now();
  const int N = (int)nodes.size();
  if (N == 0) { reset_keep_capacity(A); *msg = "empty graph"; return 0; }
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[analysis] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  auto from_scratch = [&](int why) { if (timing) fprintf(stderr, "[analysis] reuse abandoned at check %d\n", why); return 2; };
  // ---- 0. may the previous result be built upon?  The graph must be the cached one plus appended nodes / factors, every
  // new factor touching a new node (then nothing left of the new poses has a new neighbour) ----
  bool offsets_changed = false;
  if (reuse) {
    const size_t N0 = C->nodes.size(), M0 = C->factors.size();
    bool ok = C->valid && !general_ordering && prm.aligned_cuts && std::max(2, prm.arity) == 2 && N0 <= nodes.size() && M0 <= factors.size() &&
              C->prm.leaf_poses == prm.leaf_poses && C->prm.max_pivots == prm.max_pivots && C->prm.seg_len == prm.seg_len &&
              C->prm.band_levels == prm.band_levels && C->prm.band_rows == prm.band_rows && C->prm.aligned_cuts == prm.aligned_cuts &&
              C->prm.dense_min == prm.dense_min && C->prm.dense_mult == prm.dense_mult && C->prm.front_rows == prm.front_rows &&
              C->prm.cut_shift == prm.cut_shift && (int)A.f_b.size() == C->fronts_total;
    // (SymNode / SymFactor are plain ints without padding: the old parts are compared as bytes)
    static_assert(sizeof(SymNode) == 3 * sizeof(int) && sizeof(SymFactor) == 6 * sizeof(int), "byte-wise comparison of the cached graph");
    ok = ok && (N0 == 0 || memcmp(nodes.data(), C->nodes.data(), N0 * sizeof(SymNode)) == 0);
    if (ok && M0 > 0 && memcmp(factors.data(), C->factors.data(), M0 * sizeof(SymFactor)) != 0) {
      // The same factors between the same nodes at other buffer offsets (a Jacobian slab outgrew its capacity and the slabs behind it
      // moved): the tree and everything indexed by fronts is still the previous one; the H blocks and their contribution lists,
      // which hold the offsets, are listed again in full.
      for (size_t i = 0; ok && i < M0; i++) ok = factors[i].type == C->factors[i].type && factors[i].a == C->factors[i].a && factors[i].b == C->factors[i].b;
      offsets_changed = true;
    }
    for (size_t i = M0; ok && i < factors.size(); i++) ok = factors[i].a >= (int)N0 || factors[i].b >= (int)N0;
    if (!ok) return from_scratch(1);
  }
  if (!reuse) reset_keep_capacity(A);
  A.n_nodes = N;
  lap("validate / reset");
  Builder B(nodes, factors, prm);
  // (the adjacency of the previous analysis extended by the new factors, when there is one to extend)
  const bool adj_inc = reuse && C->adj_off.size() == C->nodes.size() + 1 && C->adj2_off.size() == C->nodes.size() + 1;
  if (adj_inc) {
    B.extend_adjacency(C->adj_off, C->adj, C->nodes.size(), C->factors.size(), B.adj_off, B.adj, [](int, int) { return true; });
    B.any_cross = C->any_cross || B.cross_edges_among(C->factors.size());
  } else {
    B.build_adjacency();
    B.any_cross = B.cross_edges_among(0);
  }
  lap("adjacency");
  B.lidx.assign(N, -1);
  B.dense.assign(N, 0);

  // ---- 1. dense nodes -> root border ----
  const double thr = std::max((double)prm.dense_min, prm.dense_mult * std::sqrt((double)N));
  std::vector<int> dense_nodes, poses, planes;
  for (int u = 0; u < N; u++) {
    if (B.degree(u) > thr) { B.dense[u] = 1; dense_nodes.push_back(u); }
    else if (nodes[u].type == NODE_POSE) poses.push_back(u);
    else planes.push_back(u);
  }
  for (int u : dense_nodes) B.border_dim += nodes[u].dim;
  if (reuse) {                                            // the border block must be the one it was
    for (size_t u = 0; u < C->dense.size(); u++) if (C->dense[u] != B.dense[u]) return from_scratch(2);
    for (size_t u = C->dense.size(); u < (size_t)N; u++) if (B.dense[u]) return from_scratch(3);
  }
  lap("  dense / lists");
  std::sort(poses.begin(), poses.end(), [&](int a, int b) { return nodes[a].rank < nodes[b].rank; });
  lap("  pose sort");
  // strip dense nodes from the adjacency the dissection sees
  std::vector<int> post;
  std::vector<int> f_pos0, f_npiv;
  std::vector<int> adj2_off_keep, adj2_keep;
  std::vector<char> touched;
  int root = -1;
  {
    std::vector<int> off, a2;
    if (adj_inc) {                 // (the border is the one it was: checked above)
      B.extend_adjacency(C->adj2_off, C->adj2, C->nodes.size(), C->factors.size(), off, a2, [&](int u, int v) { return !B.dense[u] && !B.dense[v]; });
    } else {
      off.assign(N + 1, 0);
      a2.reserve(B.adj.size());
      for (int u = 0; u < N; u++) {
        for (int q = B.adj_off[u]; q < B.adj_off[u + 1]; q++)
          if (!B.dense[B.adj[q]] && !B.dense[u]) a2.push_back(B.adj[q]);
        off[u + 1] = (int)a2.size();
      }
    }
    B.adj_off.swap(off);           // (off / a2 hold the full adjacency until the swap back below)
    B.adj.swap(a2);
    lap("  border stripped");
    int top = -1;
    if (!poses.empty() || !planes.empty()) {
      if (general_ordering) {
        std::vector<int> live(poses);
        live.insert(live.end(), planes.begin(), planes.end());
        top = B.mindeg_tree(live);
      } else {
        if (reuse) { C->valid = false;   // its tree and memos are moved from; set again when this analysis has succeeded
          B.old_tree = &C->tree; B.old_memo = &C->memo; B.old_memo_of_first = &C->memo_of_first; B.old_memo_next = &C->memo_next;
          touched.assign(N, 0);
          const int n_old = (int)C->nodes.size();
          for (size_t i = C->factors.size(); i < factors.size(); i++) {
            if (factors[i].a < n_old) touched[factors[i].a] = 1;
            if (factors[i].b >= 0 && factors[i].b < n_old) touched[factors[i].b] = 1;
          }
          B.touched = &touched; }
        if (!poses.empty()) { B.chain_first = poses.front(); B.chain_last = poses.back(); }
        top = B.dissect(poses, planes);
      }
    }
    B.adj_off.swap(off);
    B.adj.swap(a2);
    adj2_off_keep.swap(off); adj2_keep.swap(a2);       // (the dissection's view: what the next analysis extends)
    root = top;
    if (!dense_nodes.empty()) {
      // planes first, poses last
      std::vector<int> border;
      for (int u : dense_nodes) if (nodes[u].type == NODE_PLANE) border.push_back(u);
      for (int u : dense_nodes) if (nodes[u].type == NODE_POSE) border.push_back(u);
      // A small border block (the ground plane: 3 scalars) joins the top separator of the dissection as its last pivots instead
      // of forming a front of its own: that front's boundary is the border block anyway, so the merged front has the same rows
      // and the tree loses a level -- one front latency less on the critical path of every factorisation and back-substitution.
      int top_dim = 0, border_dim = 0;
      if (top >= 0) for (int u : B.tree[top].piv) top_dim += nodes[u].dim;
      for (int u : border) border_dim += nodes[u].dim;
      // A top taken over from the previous analysis (a re-analysis of an unchanged graph) already ends with the border: its
      // pivots must not be counted twice by the size test -- a 45-scalar separator + the 3-scalar ground plane against max_pivots
      // 48 passed the first time and would fail the second, building a separate root over nodes the top already holds.
      bool already = false;
      if (top >= 0) {
        const std::vector<int>& tp = B.tree[top].piv;
        already = tp.size() >= border.size() && std::equal(border.begin(), border.end(), tp.end() - (std::ptrdiff_t)border.size());
        if (already) top_dim -= border_dim;
      }
      if (top >= 0 && !general_ordering && top_dim + border_dim <= prm.max_pivots) {
        if (!already) B.tree[top].piv.insert(B.tree[top].piv.end(), border.begin(), border.end());
      } else if (already) {
        return from_scratch(4);                    // (a taken-over top that holds the border although the merge no longer applies: from scratch)
      } else {
        root = B.new_tnode();
        B.tree[root].piv = border;
        if (top >= 0) B.tree[root].kids.push_back(top);
      }
    }
  }
  lap("dissection");
  // ---- 2./3. post-order over the separator tree; oversized supernodes are emitted as a chain of
  // fronts (first chunk child-most: it receives the tnode's children) ----
  post.reserve(B.tree.size());
  {
    std::vector<std::pair<int, size_t>> st;
    st.emplace_back(root, 0);
    while (!st.empty()) {
      auto& top2 = st.back();
      const int t = top2.first;
      if (top2.second < B.tree[t].kids.size()) { const int c = B.tree[t].kids[top2.second++]; st.emplace_back(c, 0); }
      else { post.push_back(t); st.pop_back(); }
    }
  }
  // the tree nodes taken over from the previous analysis that also sit at the same place of the post-order: their fronts
  // keep their numbers, positions and everything indexed by them
  size_t K0 = 0;
  if (reuse) while (K0 < post.size() && K0 < C->post.size() && B.t_reused[post[K0]] && B.t_old[post[K0]] == C->post[K0]) K0++;
  A.node_voff.assign(N, -1);
  { int off = 0; for (int u = 0; u < N; u++) { A.node_voff[u] = off; off += nodes[u].dim; } }   // creation order: append-only
  std::vector<int> tn_last(B.tree.size(), -1);
  std::vector<int> post_fend(post.size(), 0);             // fronts emitted up to and including post[k]
  int pos = 0, voff = 0;
  int F0 = 0;                                             // fronts kept from the previous analysis
  // The first K0 tree nodes are the previous analysis' own, in the same places: their fronts, positions and pivot index lists are
  // what `A` still holds -- nothing of that part is written again (a frame of a frame loop redoes its right spine only).
  if (K0 > 0 && C->post_fend.size() >= K0 && (int)A.order.size() == (int)C->nodes.size() && A.node_pos.size() == C->nodes.size()) {
    F0 = C->post_fend[K0 - 1];
    if (F0 > (int)C->f_pos0.size() || F0 > (int)A.f_b.size() || F0 > (int)A.f_p.size() || F0 <= 0) return from_scratch(5);
    for (size_t k = 0; k < K0; k++) { post_fend[k] = C->post_fend[k]; tn_last[post[k]] = C->post_fend[k] - 1; }
    f_pos0.assign(C->f_pos0.begin(), C->f_pos0.begin() + F0);
    f_npiv.assign(C->f_npiv.begin(), C->f_npiv.begin() + F0);
    pos = f_pos0[F0 - 1] + f_npiv[F0 - 1];
    voff = A.f_poff[F0 - 1] + A.f_p[F0 - 1];
    for (size_t k = (size_t)pos; k < A.order.size(); k++) A.node_pos[A.order[k]] = -1;
    A.node_pos.resize(N, -1);
    A.order.resize(pos);
    A.pidx.resize(voff);
    A.f_p.resize(F0); A.f_poff.resize(F0); A.f_parent.resize(F0);
  } else {
    K0 = 0;
    A.node_pos.assign(N, -1);
    A.order.clear();
    A.pidx.clear();
    A.f_p.clear(); A.f_poff.clear(); A.f_parent.clear();
  }
  const int pos_keep = pos;                               // positions below it: nodes of the kept tree nodes, where they were
  for (size_t pi = K0; pi < post.size(); pi++) {
    const int t = post[pi];
    const TNode& tn = B.tree[t];
    // oversized supernodes are emitted as a chain of fronts: a new one starts where the next node would exceed max_pivots scalars
    int acc = 0, s = -1;
    bool first = true;
    auto open_front = [&]() {
      s = (int)A.f_p.size();
      f_pos0.push_back(pos); f_npiv.push_back(0);
      A.f_poff.push_back(voff); A.f_p.push_back(0);
      A.f_parent.push_back(-1);
      if (first) { for (int c : tn.kids) A.f_parent[tn_last[c]] = s; }
      else A.f_parent[s - 1] = s;
      first = false;
    };
    open_front();
    for (int u : tn.piv) {
      if (acc + nodes[u].dim > prm.max_pivots && acc > 0) { open_front(); acc = 0; }
      if (A.node_pos[u] != -1) { *msg = "internal: node placed twice"; return 0; }
      A.node_pos[u] = pos++;
      for (int dd = 0; dd < nodes[u].dim; dd++) A.pidx.push_back(A.node_voff[u] + dd);
      voff += nodes[u].dim;
      A.order.push_back(u);
      acc += nodes[u].dim;
      f_npiv[s] = pos - f_pos0[s];
      A.f_p[s] = voff - A.f_poff[s];
    }
    tn_last[t] = (int)A.f_p.size() - 1;
    post_fend[pi] = (int)A.f_p.size();
  }
  const int F = (int)A.f_p.size();
  A.n_fronts = F;
  if (pos != N) { *msg = "internal: ordering does not cover all nodes"; return 0; }
  A.n_scalars = voff;
  A.f_b.resize(F, 0); A.f_level.assign(F, 0);

  lap("post-order / chains");
  // ---- 4. boundaries ----
  std::vector<std::vector<int>> bnd(F);
  // children CSR (ascending inside a parent), from f_parent
  auto build_children = [&]() {
    A.f_child_off.assign(F + 1, 0);
    for (int s = 0; s < F; s++) if (A.f_parent[s] >= 0) A.f_child_off[A.f_parent[s] + 1]++;
    for (int s = 0; s < F; s++) A.f_child_off[s + 1] += A.f_child_off[s];
    A.child.resize(A.f_child_off[F]);
    std::vector<int> w(A.f_child_off.begin(), A.f_child_off.end() - 1);
    for (int s = 0; s < F; s++) if (A.f_parent[s] >= 0) A.child[w[A.f_parent[s]]++] = s;
  };
  build_children();
  struct KidRange { const int* b; const int* e; const int* begin() const { return b; } const int* end() const { return e; } };
  auto kids_of = [&](int s) { return KidRange{A.child.data() + A.f_child_off[s], A.child.data() + A.f_child_off[s + 1]}; };
  for (int s = 0; s < F0; s++) {
    bnd[s].swap(C->bnd[s]);
    // The boundary nodes of a kept front lie in later fronts, kept or redone; they must still come in the order they had
    // (a plane that moved to another separator of the redone spine changes the local layout of the fronts it bounds):
    // the kept part ends at the first front for which that no longer holds.
    // (The list is sorted by the OLD positions: the nodes of kept positions come first and have not moved; only its tail -- the
    // nodes of redone positions, the border and a few separators of the spine -- can have changed order.)
    bool same = true;
    const std::vector<int>& b = bnd[s];
    for (size_t k = b.size(); same && k > 0 && A.node_pos[b[k - 1]] >= pos_keep; k--)
      if (k < b.size()) same = A.node_pos[b[k - 1]] < A.node_pos[b[k]];
    if (!same) { F0 = s; break; }
  }
  if (reuse) C->valid = false;                            // its boundary lists are gone: from here on a failure means "from scratch"
  const int F0b = offsets_changed ? 0 : F0;               // fronts whose H blocks / contribution lists / assembly lists are kept
  const int P0 = F0b < F ? f_pos0[F0b] : N;               // positions below P0 belong to fronts whose blocks are kept
  auto compute_boundaries = [&](int s_begin) {
    std::vector<int> stamp(N, -1);
    for (int s = s_begin; s < F; s++) {
      const int end = f_pos0[s] + f_npiv[s];
      std::vector<int>& b = bnd[s];
      b.clear();
      for (int k = f_pos0[s]; k < end; k++) {
        const int u = A.order[k];
        for (int q = B.adj_off[u]; q < B.adj_off[u + 1]; q++) {
          const int v = B.adj[q];
          if (A.node_pos[v] >= end && stamp[v] != s) { stamp[v] = s; b.push_back(v); }
        }
      }
      for (int c : kids_of(s))
        for (int v : bnd[c])
          if (A.node_pos[v] >= end && stamp[v] != s) { stamp[v] = s; b.push_back(v); }
      std::sort(b.begin(), b.end(), [&](int x, int y) { return A.node_pos[x] < A.node_pos[y]; });
    }
  };
  lap("  kept boundaries");
  compute_boundaries(F0);
  lap("  new boundaries");
  // verify the separator property: every boundary node of s is a pivot of an ancestor of s,
  // and every boundary node of a child is inside the parent's front.
  std::vector<int> node_front(N);
  for (int s = 0; s < F; s++)
    for (int k = f_pos0[s]; k < f_pos0[s] + f_npiv[s]; k++) node_front[A.order[k]] = s;
  bool valid = true;
  {
    std::vector<int> anc_stamp(F, -1);
    for (int s = F0; s < F && valid; s++) {
      for (int x = A.f_parent[s]; x >= 0; x = A.f_parent[x]) anc_stamp[x] = s;
      for (int v : bnd[s]) if (anc_stamp[node_front[v]] != s) { valid = false; break; }
    }
  }
  if (!valid) {
    if (reuse) return from_scratch(6);
    // fall back to a chain: every later front is an ancestor, which is always a valid assembly tree
    for (int s = 0; s < F; s++) A.f_parent[s] = (s + 1 < F) ? s + 1 : -1;
    build_children();
    compute_boundaries(0);
  }
  lap("  separator check");
  // levels
  A.n_levels = 0;
  for (int s = 0; s < F; s++) {
    int lv = 0;
    for (int c : kids_of(s)) lv = std::max(lv, A.f_level[c] + 1);
    A.f_level[s] = lv;
    A.n_levels = std::max(A.n_levels, lv + 1);
  }
  A.level_off.assign(A.n_levels + 1, 0);
  for (int s = 0; s < F; s++) A.level_off[A.f_level[s] + 1]++;
  for (int l = 0; l < A.n_levels; l++) A.level_off[l + 1] += A.level_off[l];
  A.level_fronts.resize(F);
  {
    std::vector<int> w(A.level_off.begin(), A.level_off.end() - 1);
    for (int s = 0; s < F; s++) A.level_fronts[w[A.f_level[s]]++] = s;
  }
  lap("boundaries");
  // ---- band schedule ----
  {
    // levels per band.  Given, or (0) chosen here from what the tree turned out to be -- measured on one MI355X, us per LM iteration, corridor
    // graphs (tools/size_probe.py; builds with a fixed depth): up to 2 250 poses three levels per band (1 000 poses 57.8 against 62.7 with four,
    // 2 000 poses 75.2 / 78.1, 2 250 poses 84.2 / 88.8); from there on FOUR (2 500 poses 108.7 -> 95.9, 3 000 101.7 -> 87.8, 3 900 99.9 -> 87.3, and
    // against the two levels of rounds 3 - 5: 4 000 poses 107.5 -> 91.4, 6 000 142.9 -> 128.4, 8 000 140.3 -> 124.4, 12 000 184.8 -> 172.1) --
    // unless a front needs the fifteen-tile kernel, which a band of four levels would put under more fronts (C3: 357 us with two, 369 with four).
    int Bn = std::max(0, prm.band_levels);
    if (Bn == 0) {
      int n_poses = 0, widest = 0;
      for (int u = 0; u < N; u++) n_poses += nodes[u].type == NODE_POSE ? 1 : 0;
      for (int s = 0; s < F; s++) {
        int rows = A.f_p[s];
        for (int v : bnd[s]) rows += nodes[v].dim;
        widest = std::max(widest, rows);
      }
      Bn = n_poses < 2400 ? 3 : (prm.front_rows > 0 && widest > prm.front_rows ? 2 : 4);
    }
    A.band_levels = Bn;
    A.n_stages = (A.n_levels + Bn - 1) / Bn;
    std::vector<int> grp(F, -1), ll(F, 0);
    std::vector<int> grp_stage;
    for (int s = F - 1; s >= 0; s--) {       // parents before children (post-order reversed)
      const int par = A.f_parent[s];
      const int band = A.f_level[s] / Bn;
      if (par < 0 || A.f_level[par] / Bn != band) { grp[s] = (int)grp_stage.size(); grp_stage.push_back(band); }
      else grp[s] = grp[par];
    }
    for (int s = 0; s < F; s++) {             // children before parents
      int l = 0;
      for (int c : kids_of(s)) if (grp[c] == grp[s]) l = std::max(l, ll[c] + 1);
      ll[s] = l;
    }
    const int G = (int)grp_stage.size();
    // order groups by stage
    std::vector<int> gorder(G), gnew(G);
    std::iota(gorder.begin(), gorder.end(), 0);
    std::stable_sort(gorder.begin(), gorder.end(), [&](int a2, int b2) { return grp_stage[a2] < grp_stage[b2]; });
    for (int i = 0; i < G; i++) gnew[gorder[i]] = i;
    A.n_groups = G;
    A.stage_grp_off.assign(A.n_stages + 1, 0);
    for (int g2 = 0; g2 < G; g2++) A.stage_grp_off[grp_stage[g2] + 1]++;
    for (int st = 0; st < A.n_stages; st++) A.stage_grp_off[st + 1] += A.stage_grp_off[st];
    std::vector<int> g_nl(G, 0);
    for (int s = 0; s < F; s++) g_nl[gnew[grp[s]]] = std::max(g_nl[gnew[grp[s]]], ll[s] + 1);
    A.grp_lvl_off.assign(G + 1, 0);
    for (int g2 = 0; g2 < G; g2++) A.grp_lvl_off[g2 + 1] = A.grp_lvl_off[g2] + g_nl[g2];
    A.n_glevels = A.grp_lvl_off[G];
    A.glvl_front_off.assign(A.n_glevels + 1, 0);
    for (int s = 0; s < F; s++) A.glvl_front_off[A.grp_lvl_off[gnew[grp[s]]] + ll[s] + 1]++;
    for (int i = 0; i < A.n_glevels; i++) A.glvl_front_off[i + 1] += A.glvl_front_off[i];
    A.glvl_fronts.assign(F, 0);
    std::vector<int> w(A.glvl_front_off.begin(), A.glvl_front_off.end() - 1);
    for (int s = 0; s < F; s++) A.glvl_fronts[w[A.grp_lvl_off[gnew[grp[s]]] + ll[s]]++] = s;
    A.stage_max_front.assign(A.n_stages, 0);
    A.stage_max_width.assign(A.n_stages, 0);
    for (int g2 = 0; g2 < G; g2++) {
      int st = 0;
      while (g2 >= A.stage_grp_off[st + 1]) st++;
      for (int l = A.grp_lvl_off[g2]; l < A.grp_lvl_off[g2 + 1]; l++)
        A.stage_max_width[st] = std::max(A.stage_max_width[st], A.glvl_front_off[l + 1] - A.glvl_front_off[l]);
    }
  }
  // boundary scalar indices, sizes, storage: the kept fronts keep theirs (every offset is cumulative in front order)
  const int bidx0 = F0 > 0 ? A.f_bidx_off[F0] : 0;
  A.L_size = F0 > 0 ? (F0 < (int)A.f_Loff.size() ? A.f_Loff[F0] : A.L_size) : 0;
  A.U_size = F0 > 0 ? (F0 < (int)A.f_Uoff.size() ? A.f_Uoff[F0] : A.U_size) : 0;
  A.f_bidx_off.resize(F + 1, 0);
  A.bidx.resize(bidx0);
  A.f_Loff.resize(F, 0); A.f_Uoff.resize(F, 0);
  if (F0 == 0) A.f_bidx_off[0] = 0;
  for (int s = F0; s < F; s++) {
    int b = 0;
    for (int v : bnd[s]) { for (int d = 0; d < nodes[v].dim; d++) A.bidx.push_back(A.node_voff[v] + d); b += nodes[v].dim; }
    A.f_b[s] = b;
    A.f_bidx_off[s + 1] = (int)A.bidx.size();
    const int f = A.f_p[s] + b;
    A.f_Loff[s] = A.L_size; A.L_size += (int64_t)(f + 1) * A.f_p[s];
    A.f_Uoff[s] = A.U_size; A.U_size += (int64_t)(b + 1) * (b + 1);
  }
  A.max_front = 0;
  for (int s = 0; s < F; s++) {
    A.max_front = std::max(A.max_front, A.f_p[s] + A.f_b[s]);
    const int st = A.f_level[s] / std::max(1, A.band_levels);
    A.stage_max_front[st] = std::max(A.stage_max_front[st], A.f_p[s] + A.f_b[s]);
  }
  // child -> parent scatter maps: a kept front whose parent is kept as well keeps its map; the children of redone fronts
  // (kept or not) get theirs from the parent's new layout -- same length, written in place
  std::vector<int> loc(N, -1);
  {
    std::vector<int> cm_off(F + 1, 0);
    for (int s = 0; s < F; s++) cm_off[s + 1] = cm_off[s] + (A.f_parent[s] >= 0 ? A.f_b[s] + 1 : 0);
    if (F0 > 0) for (int s = 0; s <= F0; s++) if (cm_off[s] != A.f_cmap_off[s]) return from_scratch(7);
    A.f_cmap_off = cm_off;
    A.cmap.resize(cm_off[F]);
    for (int s = F0; s < F; s++) {
      int off = 0;
      for (int k = f_pos0[s]; k < f_pos0[s] + f_npiv[s]; k++) { loc[A.order[k]] = off; off += nodes[A.order[k]].dim; }
      for (int v : bnd[s]) { loc[v] = off; off += nodes[v].dim; }
      const int rhs_row = off;   // == f
      for (int c : kids_of(s)) {
        int* out = A.cmap.data() + cm_off[c];
        for (int v : bnd[c]) {
          if (loc[v] < 0) { *msg = "internal: child boundary not inside parent front"; return 0; }
          for (int d = 0; d < nodes[v].dim; d++) *out++ = loc[v] + d;
        }
        *out++ = rhs_row;
        if (out != A.cmap.data() + cm_off[c + 1]) { *msg = "internal: child map length"; return 0; }
      }
      for (int k = f_pos0[s]; k < f_pos0[s] + f_npiv[s]; k++) loc[A.order[k]] = -1;
      for (int v : bnd[s]) loc[v] = -1;
    }
  }
  // packed update matrix of c -> packed index in the parent front
  // (only the wave-per-front kernels read these lists; a graph with wider fronts runs in the dense-front form, which
  // pulls through cmap -- its (b+1)^2/2 entries per front would be gigabytes for loop-closure separators)
  A.f_ea_off.assign(F + 1, 0);
  A.ea_tgt.clear();
  // Only the offsets are computed here: the lists themselves (sum of (b+1)(b+2)/2 entries, ~0.3 M for a
  // 1000-pose graph) are expanded from cmap on the device (k_expand_ea) -- or by expand_ea_tgt() for the dump.
  A.ea_total = 0;
  if (A.max_front <= prm.band_rows) {
    for (int s = 0; s < F; s++) {
      A.f_ea_off[s] = A.ea_total;
      const int64_t len = A.f_cmap_off[s + 1] - A.f_cmap_off[s];             // the root has no map
      A.ea_total += len * (len + 1) / 2;
    }
    A.f_ea_off[F] = A.ea_total;
  }

  lap("band schedule");
  // ---- 5. block-sparse H and contribution lists ----
  // Blocks are numbered by (column position, row position); the blocks whose column is a kept position, with their
  // segments and contribution lists, are the ones of the previous analysis.
  struct Ctr { int pv, pu, jv, ju, roff, m, fi; };   // (row position, column position) of the H block, the J slices, the factor
  int B0 = 0;                                          // blocks kept
  if (F0b > 0) B0 = (int)(std::lower_bound(C->blk_pu.begin(), C->blk_pu.end(), P0) - C->blk_pu.begin());
  const int S0 = B0 > 0 ? (B0 < (int)A.blk_hoff.size() ? (int)(std::lower_bound(A.seg_blk.begin(), A.seg_blk.end(), B0) - A.seg_blk.begin()) : (int)A.seg_blk.size()) : 0;
  const int C0 = S0 > 0 ? (S0 < (int)A.seg_c0.size() ? A.seg_c0[S0] : (int)(A.contrib.size() / 4)) : 0;
  const int64_t H0 = B0 > 0 ? (B0 < (int)A.blk_hoff.size() ? A.blk_hoff[B0] : A.H_size) : 0;
  std::vector<Ctr> ctr;
  // The factors of every node, in factor order (append-only: the next analysis of the grown graph adds the new factors).  The
  // contributions are listed COLUMN BY COLUMN from them -- only the columns of the redone positions are visited, so a frame of a
  // frame loop walks the factors of its right spine and of the border, not every factor of the graph.
  std::vector<std::vector<int>> inc_local;
  std::vector<std::vector<int>>& inc = C ? C->inc : inc_local;
  size_t m_from = 0;
  if (reuse && !offsets_changed && C->inc_factors <= factors.size() && C->inc.size() <= (size_t)N && C->inc_factors == C->factors.size()) {
    m_from = C->inc_factors;
    A.J_size = C->J_size; A.P_size = C->P_size;
  } else {
    for (auto& v : inc) v.clear();
    A.J_size = 0; A.P_size = 0;
    if (C) C->n_obs_slots = 0;
  }
  inc.resize(N);
  int n_obs_slots = (C && m_from > 0) ? C->n_obs_slots : 0;
  for (size_t fi2 = m_from; fi2 < factors.size(); fi2++) {
    const auto& f = factors[fi2];
    if (f.type == F_PLANE_OBS) n_obs_slots = std::max(n_obs_slots, f.joff / kJSize[F_PLANE_OBS] + 1);
    const int m = kFDim[f.type];
    const int da = nodes[f.a].dim;
    const int db = f.b >= 0 ? nodes[f.b].dim : 0;
    A.J_size = std::max<int64_t>(A.J_size, (int64_t)f.joff + m * (da + db) + m);
    if (f.type == F_PLANE_OBS && f.direct_ok) A.P_size = std::max<int64_t>(A.P_size, (int64_t)f.poff + kPSize[f.type]);
    inc[f.a].push_back((int)fi2);
    if (f.b >= 0 && f.b != f.a) inc[f.b].push_back((int)fi2);
  }
  if (C) { C->inc_factors = factors.size(); C->J_size = A.J_size; C->P_size = A.P_size; C->n_obs_slots = n_obs_slots; }
  // Column p2 holds: the diagonal block of its node (one contribution per factor of the node), and the off-diagonal blocks towards
  // the nodes eliminated later (rows = the later node).  Inside a column the contributions come in factor order; a stable insertion
  // sort on the row position then groups the blocks -- (column, row) ascending, factor order inside a block.
  // A plain plane observation (direct_ok) hands K2 its PRODUCT record instead of its Jacobian for the two diagonal blocks it
  // feeds: the contribution carries the record's offset in `ju` and kProductFlag on top of the row count.
  for (int p2 = P0; p2 < N; p2++) {
    const int u = A.order[p2];
    const size_t c_begin = ctr.size();
    for (const int fi : inc[u]) {
      const auto& f = factors[fi];
      const int m = kFDim[f.type];
      const int da = nodes[f.a].dim;
      const int db = f.b >= 0 ? nodes[f.b].dim : 0;
      const int ja = f.joff, jb = f.joff + m * da, roff = f.joff + m * (da + db);
      const bool prod = f.type == F_PLANE_OBS && f.direct_ok;
      if (f.a == u) {
        ctr.push_back({p2, p2, ja, prod ? f.poff : ja, roff, prod ? m + kProductFlag : m, fi});
        if (f.b >= 0 && f.b != f.a) { const int pb = A.node_pos[f.b]; if (pb >= p2) ctr.push_back({pb, p2, jb, ja, roff, m, fi}); }   // rows = later node
      } else {
        ctr.push_back({p2, p2, jb, prod ? f.poff + da * da + da : jb, roff, prod ? m + kProductFlag : m, fi});
        const int pa = A.node_pos[f.a];
        if (pa > p2) ctr.push_back({pa, p2, ja, jb, roff, m, fi});
      }
    }
    // every node needs a diagonal block even when it has no factor (it will then fail as not PD)
    if (inc[u].empty()) ctr.push_back({p2, p2, 0, 0, 0, 0, -1});
    Ctr* b = ctr.data() + c_begin;
    const int n = (int)(ctr.size() - c_begin);
    if (n > 64) {                // (the border's column: a thousand contributions to its diagonal block, in order as they come)
      const auto by_row = [](const Ctr& x, const Ctr& y) { return x.pv < y.pv; };
      if (!std::is_sorted(b, b + n, by_row)) std::stable_sort(b, b + n, by_row);
      continue;
    }
    for (int i = 1; i < n; i++) {
      const Ctr c = b[i];
      int j = i - 1;
      while (j >= 0 && b[j].pv > c.pv) { b[j + 1] = b[j]; j--; }
      b[j + 1] = c;
    }
  }
  lap("  contributions sorted");
  A.contrib.resize((size_t)C0 * 4);
  // (no exact-size reserve here: it would re-allocate and copy the whole list on every frame of a frame loop; push_back grows
  // geometrically)
  if (offsets_changed) A.obs_dir.clear();                 // (slots are Jacobian offsets)
  A.obs_dir.resize(3 * (size_t)n_obs_slots, -1);
  // an observation whose (pose, plane) block is redone starts as "not direct"
  for (const auto& c : ctr)
    if (c.pv != c.pu && c.fi >= 0 && factors[c.fi].direct_ok) A.obs_dir[3 * (size_t)(factors[c.fi].joff / kJSize[F_PLANE_OBS])] = -1;
  A.nd_segs.resize((size_t)(std::lower_bound(A.nd_segs.begin(), A.nd_segs.end(), S0) - A.nd_segs.begin()));      // (ascending)
  const int nd_kept = (int)A.nd_segs.size();
  A.blk_rows.resize(B0); A.blk_cols.resize(B0); A.blk_size.resize(B0); A.blk_nseg.resize(B0); A.blk_hoff.resize(B0);
  A.seg_blk.resize(S0); A.seg_c0.resize(S0); A.seg_cnt.resize(S0); A.seg_hoff.resize(S0);
  A.n_blocks = B0; A.n_segs = S0; A.H_size = H0;
  lap("   resizes");
  std::vector<int> blk_pu;                             // column position per block (kept part from the cache)
  if (B0 > 0) blk_pu.assign(C->blk_pu.begin(), C->blk_pu.begin() + B0);
  std::vector<std::vector<int>> asm_of(F);             // block ids per (redone) front
  lap("   blk_pu / asm_of");
  std::vector<int> blk_v, blk_u;                       // row / column node of the redone blocks (index blk - B0)
  std::vector<int> blk_is_direct_slot;                 // redone block -> observation slot when direct, else -1
  for (size_t i = 0; i < ctr.size();) {
    size_t j = i;
    while (j < ctr.size() && ctr[j].pv == ctr[i].pv && ctr[j].pu == ctr[i].pu) j++;
    const int pv = ctr[i].pv, pu = ctr[i].pu;
    const int v = A.order[pv], u = A.order[pu];
    const int rows = nodes[v].dim, cols = nodes[u].dim;
    const int size = rows * cols + (v == u ? rows : 0);
    const int cnt = (int)(j - i);
    const int nseg = std::max(1, (cnt + prm.seg_len - 1) / prm.seg_len);
    const int blk = A.n_blocks++;
    A.blk_rows.push_back(rows); A.blk_cols.push_back(cols); A.blk_size.push_back(size); A.blk_nseg.push_back(nseg);
    A.blk_hoff.push_back(A.H_size);
    blk_pu.push_back(pu);
    // a (pose, plane) block fed by exactly one plain plane observation: K1 may write it (see pps_symbolic.h)
    const bool direct = v != u && cnt == 1 && ctr[i].fi >= 0 && factors[ctr[i].fi].direct_ok;
    if (direct) {
      const int slot = factors[ctr[i].fi].joff / kJSize[F_PLANE_OBS];
      A.obs_dir[3 * (size_t)slot + 0] = (int)A.H_size;
      A.obs_dir[3 * (size_t)slot + 1] = -1;                           // the block's Hf offset: set below
      A.obs_dir[3 * (size_t)slot + 2] = nodes[v].type == NODE_POSE ? 1 : 0;
      blk_is_direct_slot.push_back(slot);
    } else {
      for (int sgi = 0; sgi < nseg; sgi++) A.nd_segs.push_back(A.n_segs + sgi);
      blk_is_direct_slot.push_back(-1);
    }
    for (int sgi = 0; sgi < nseg; sgi++) {
      const int c0 = sgi * prm.seg_len;
      A.seg_blk.push_back(blk);
      A.seg_c0.push_back((int)(A.contrib.size() / 4));
      int k = 0;
      for (; k < prm.seg_len && c0 + k < cnt; k++) {
        const Ctr& c = ctr[i + c0 + k];
        if (c.m == 0) continue;   // placeholder for a factor-less node
        A.contrib.push_back(c.jv); A.contrib.push_back(c.ju); A.contrib.push_back(c.roff); A.contrib.push_back(c.m);
      }
      A.seg_cnt.push_back((int)(A.contrib.size() / 4) - A.seg_c0.back());
      A.seg_hoff.push_back(A.H_size);
      A.H_size += size;
      A.n_segs++;
    }
    asm_of[node_front[u]].push_back(blk);
    blk_v.push_back(v); blk_u.push_back(u);
    i = j;
  }
  lap("  blocks / segments");
  // per-front assembly lists with local offsets
  const int asm0 = F0b > 0 ? A.f_asm_off[F0b] : 0;
  A.el_total = F0b > 0 ? A.f_el_off[F0b] : 0;
  A.f_asm_off.resize(F + 1, 0); A.f_el_off.resize(F + 1, 0);
  if (F0b == 0) { A.f_asm_off[0] = 0; A.f_el_off[0] = 0; }
  A.asm_blk.resize(asm0); A.asm_lrow.resize(asm0); A.asm_lcol.resize(asm0); A.asm_el0.resize(asm0); A.asm_fsz.resize(asm0);
  A.el_tgt.clear();
  A.blk_doff.resize(A.n_blocks + 1, 0);
  if (B0 == 0) A.blk_doff[0] = 0;
  for (int bk = B0; bk < A.n_blocks; bk++) A.blk_doff[bk + 1] = A.blk_doff[bk] + A.blk_size[bk];
  A.blk_dst.clear();
  if (A.H_size > 0x3fffffffLL) { *msg = "H too large for int32 gather offsets"; return 0; }
  for (int s = F0b; s < F; s++) {
    int off = 0;
    for (int k = f_pos0[s]; k < f_pos0[s] + f_npiv[s]; k++) { loc[A.order[k]] = off; off += nodes[A.order[k]].dim; }
    for (int v : bnd[s]) { loc[v] = off; off += nodes[v].dim; }
    const int fsz = A.f_p[s] + A.f_b[s];
    for (int blk : asm_of[s]) {
      const int v = blk_v[blk - B0], u = blk_u[blk - B0];
      if (loc[v] < 0 || loc[u] < 0) { *msg = "internal: H block outside its front"; return 0; }
      A.asm_blk.push_back(blk); A.asm_lrow.push_back(loc[v]); A.asm_lcol.push_back(loc[u]);
      // flat element list of the front (lower triangle of diagonal blocks, full off-diagonal blocks, g entries): only
      // where each assembled block starts is recorded here; el_tgt / blk_dst are expanded from that on the device
      // (k_expand_el) or by expand_el_lists() for the dump
      const int rows = A.blk_rows[blk], cols = A.blk_cols[blk];
      const bool diag = v == u;
      A.asm_el0.push_back((int)A.el_total);
      if (blk_is_direct_slot[blk - B0] >= 0) A.obs_dir[3 * (size_t)blk_is_direct_slot[blk - B0] + 1] = (int)A.el_total;
      A.asm_fsz.push_back(fsz);
      A.el_total += diag ? rows * (rows + 1) / 2 + rows : rows * cols;
    }
    A.f_asm_off[s + 1] = (int)A.asm_blk.size();
    A.f_el_off[s + 1] = (int)A.el_total;
    for (int k = f_pos0[s]; k < f_pos0[s] + f_npiv[s]; k++) loc[A.order[k]] = -1;
    for (int v : bnd[s]) loc[v] = -1;
  }
  lap("H blocks / lists");
  // ---- packed records ----
  {
    A.frec.assign((size_t)F * 16, 0);
    A.crec.clear();
    std::vector<int> pos_of(F, -1);
    for (int i = 0; i < F; i++) pos_of[A.glvl_fronts[i]] = i;
    std::vector<int> grp_first(F, 0);                    // position -> first position of its group
    for (int g2 = 0; g2 < A.n_groups; g2++) {
      const int i0 = A.glvl_front_off[A.grp_lvl_off[g2]], i1 = A.glvl_front_off[A.grp_lvl_off[g2 + 1]];
      for (int i = i0; i < i1; i++) grp_first[i] = i0;
    }
    for (int i = 0; i < F; i++) {
      const int s2 = A.glvl_fronts[i];
      int* r = &A.frec[(size_t)i * 16];
      r[0] = s2; r[1] = A.f_p[s2]; r[2] = A.f_b[s2]; r[3] = A.f_el_off[s2]; r[4] = A.f_el_off[s2 + 1];
      r[5] = (int)(A.crec.size() / 8); r[6] = A.f_child_off[s2 + 1] - A.f_child_off[s2];
      r[7] = A.f_poff[s2]; r[8] = A.f_bidx_off[s2];
      r[9] = (int)(A.f_Loff[s2] & 0xffffffffLL); r[10] = (int)(A.f_Loff[s2] >> 32);
      r[11] = (int)(A.f_Uoff[s2] & 0xffffffffLL); r[12] = (int)(A.f_Uoff[s2] >> 32);
      r[13] = -1;                                             // the parent's front when another band group holds it (hand-over between workgroups: XGroup, pps_k3.hip)
      // solve hand-down: a front whose parent sits in the same group reads its boundary values from the parent's
      // local solution vector (LDS) through cmap instead of gathering them from delta
      r[14] = -1;
      r[15] = A.f_cmap_off[s2];
      {
        const int par = A.f_parent[s2];
        const int Bn2 = std::max(1, A.band_levels);
        // slot = position inside the group (groups are contiguous in glvl_fronts: grp_first = first position of the group's
        // first local level)
        if (par >= 0 && A.f_level[par] / Bn2 == A.f_level[s2] / Bn2) r[14] = pos_of[par] - grp_first[pos_of[par]];
        else if (par >= 0) r[13] = par;
      }
      for (int ci = A.f_child_off[s2]; ci < A.f_child_off[s2 + 1]; ci++) {
        const int c = A.child[ci];
        const int bc1 = A.f_b[c] + 1;
        int cr[8] = {bc1 * (bc1 + 1) / 2, (int)(A.f_Uoff[c] & 0xffffffffLL), (int)(A.f_Uoff[c] >> 32),
                     (int)(A.f_ea_off[c] & 0xffffffffLL), (int)(A.f_ea_off[c] >> 32), c,
                     A.f_level[c] / std::max(1, A.band_levels) != A.f_level[s2] / std::max(1, A.band_levels) ? 1 : 0, 0};      // slot 6: the child belongs to another band group
        A.crec.insert(A.crec.end(), cr, cr + 8);
      }
    }
    {
      A.cls_off.assign((size_t)A.n_levels * 3 + 1, 0);
      A.cls_fronts.assign(F, 0);
      // size class of a front = tile rows of the level kernel that takes it (kb_level_factor2 / 3 / 4).  A class that holds a handful of a
      // level's fronts next to sixteen times as many of the next class is merged into that class: a launch of its own costs its latency
      // (10 us for four fronts of a C2 tree's leaf level), the larger kernel runs a smaller front at the price of its empty tiles.
      std::vector<int> raw((size_t)A.n_levels * 3, 0);
      auto rows_cls = [&](int s2) { const int f = A.f_p[s2] + A.f_b[s2]; return f <= 32 ? 0 : (f <= 48 ? 1 : 2); };
      for (int s2 = 0; s2 < F; s2++) raw[(size_t)A.f_level[s2] * 3 + rows_cls(s2)]++;
      std::vector<signed char> up((size_t)A.n_levels * 3, 0);          // class c of level l runs as class up[3 l + c]
      for (int l = 0; l < A.n_levels; l++) {
        int* r = &raw[(size_t)l * 3];
        signed char* u = &up[(size_t)l * 3];
        u[0] = 0; u[1] = 1; u[2] = 2;
        if (r[1] > 0 && r[1] * 16 <= r[2]) { u[1] = 2; r[2] += r[1]; r[1] = 0; }
        if (r[0] > 0) {
          if (r[1] > 0 && r[0] * 16 <= r[1]) u[0] = 1;
          else if (r[1] == 0 && r[2] > 0 && r[0] * 16 <= r[2]) u[0] = 2;
        }
      }
      auto cls_of = [&](int s2) { return (int)up[(size_t)A.f_level[s2] * 3 + rows_cls(s2)]; };
      for (int i = 0; i < F; i++) { const int s2 = A.glvl_fronts[i]; A.cls_off[(size_t)A.f_level[s2] * 3 + cls_of(s2) + 1]++; }
      for (size_t k = 0; k + 1 < A.cls_off.size(); k++) A.cls_off[k + 1] += A.cls_off[k];
      std::vector<int> w(A.cls_off.begin(), A.cls_off.end() - 1);
      for (int i = 0; i < F; i++) { const int s2 = A.glvl_fronts[i]; A.cls_fronts[w[(size_t)A.f_level[s2] * 3 + cls_of(s2)]++] = i; }
    }
    A.srec.resize((size_t)A.n_segs * 8, 0);
    for (int sg = S0; sg < A.n_segs; sg++) {
      const int bk = A.seg_blk[sg];
      int* r = &A.srec[(size_t)sg * 8];
      r[0] = A.blk_rows[bk]; r[1] = A.blk_cols[bk]; r[2] = A.blk_size[bk]; r[3] = A.seg_c0[sg]; r[4] = A.seg_cnt[sg];
      r[5] = (int)A.seg_hoff[sg]; r[6] = A.blk_doff[bk]; r[7] = A.blk_nseg[bk];
    }
    lap("packed records");
  }
  A.kept.fronts = reuse ? F0 : 0; A.kept.fronts_lists = reuse ? F0b : 0; A.kept.blocks = reuse ? B0 : 0; A.kept.segs = reuse ? S0 : 0;
  A.kept.contribs = reuse ? C0 : 0; A.kept.nd_segs = reuse ? nd_kept : 0;
  // ---- what the next analysis of this graph may build upon ----
  if (C && !general_ordering) {
    if (reuse && !offsets_changed && C->nodes.size() <= nodes.size() && C->factors.size() <= factors.size()) {      // (the old parts were found equal above)
      C->nodes.insert(C->nodes.end(), nodes.begin() + C->nodes.size(), nodes.end());
      C->factors.insert(C->factors.end(), factors.begin() + C->factors.size(), factors.end());
    } else { C->nodes = nodes; C->factors = factors; }
    C->prm = prm;
    C->dense = B.dense;
    C->tree.swap(B.tree);
    C->memo.swap(B.memo);
    C->memo_of_first.assign(N, -1);
    C->memo_next.assign(C->memo.size(), -1);
    for (int mi = (int)C->memo.size() - 1; mi >= 0; mi--) {
      const int fp = C->memo[mi].first;
      if (fp < 0) continue;
      C->memo_next[mi] = C->memo_of_first[fp];
      C->memo_of_first[fp] = mi;
    }
    C->post.swap(post);
    C->post_fend.swap(post_fend);
    C->f_pos0.swap(f_pos0); C->f_npiv.swap(f_npiv);
    C->bnd.swap(bnd);
    C->blk_pu.swap(blk_pu);
    C->adj_off.swap(B.adj_off); C->adj.swap(B.adj);
    C->adj2_off.swap(adj2_off_keep); C->adj2.swap(adj2_keep);
    C->any_cross = B.any_cross;
    C->fronts_reused = F0; C->fronts_total = F;
    C->valid = valid;                                    // (a chain fall-back is not something to build upon)
  } else if (C) {
    C->valid = false;
  }
  lap("cache");
  return 1;
}

double factor_flops(const Analysis& A) {
  double fl = 0;
  for (int s = 0; s < A.n_fronts; s++) { const double f = A.f_p[s] + A.f_b[s] + 1.0; fl += A.f_p[s] * f * f; }
  return fl;
}
}  // namespace

// Ordering choice: the pose-chain dissection is the natural backbone of plane-SLAM graphs (fronts of a few dozen rows);
// when it leaves fronts beyond the wave-per-front kernels (pose graphs with many loop closures, 2-D meshes) the
// general minimum-degree ordering is analysed as well and the cheaper factorisation (flops) wins.
bool analyze(const std::vector<SymNode>& nodes, const std::vector<SymFactor>& factors, const AnalysisParams& prm,
             Analysis& A, const char** msg, AnalysisCache* cache) {
  int rc = 2;
  if (cache && cache->valid && prm.ordering != 1 && prm.aligned_cuts) rc = analyze_with(nodes, factors, prm, A, msg, false, cache, true);
  if (rc == 2) {
    if (cache) cache->valid = false;
    rc = analyze_with(nodes, factors, prm, A, msg, prm.ordering == 1, cache, false);
  }
  if (rc != 1) { if (cache) cache->valid = false; return false; }
  if (prm.ordering == 0 && A.max_front > prm.band_rows) {
    Analysis G;
    const char* m2 = "";
    if (analyze_with(nodes, factors, prm, G, &m2, true, nullptr, false) == 1 && factor_flops(G) < factor_flops(A)) {
      A = std::move(G);
      if (cache) cache->valid = false;
    }
  }
  return true;
}

void expand_el_lists(Analysis& A) {
  auto tri = [](int i) { return i * (i + 1) / 2; };
  A.el_tgt.assign((size_t)A.el_total, 0);
  A.blk_dst.assign((size_t)A.blk_doff[A.n_blocks], -1);
  for (size_t a = 0; a < A.asm_blk.size(); a++) {
    const int blk = A.asm_blk[a], rows = A.blk_rows[blk], cols = A.blk_cols[blk];
    const int lrow = A.asm_lrow[a], lcol = A.asm_lcol[a];
    const bool diag = A.blk_size[blk] != rows * cols;
    int e = A.asm_el0[a];
    for (int i = 0; i < rows; i++)
      for (int j = 0; j < cols; j++) {
        if (diag && j > i) continue;
        A.blk_dst[A.blk_doff[blk] + i * cols + j] = e;
        A.el_tgt[e++] = (tri(lrow + i) + lcol + j) | ((diag && i == j) ? (1 << 30) : 0);
      }
    if (diag)
      for (int i = 0; i < rows; i++) {
        A.blk_dst[A.blk_doff[blk] + rows * cols + i] = e;
        A.el_tgt[e++] = tri(A.asm_fsz[a]) + lcol + i;
      }
  }
}

void expand_ea_tgt(Analysis& A) {
  A.ea_tgt.clear();
  if (A.ea_total <= 0) return;
  A.ea_tgt.resize((size_t)A.ea_total);
  for (int s = 0; s < A.n_fronts; s++) {
    const int* m = A.cmap.data() + A.f_cmap_off[s];
    const int len = A.f_cmap_off[s + 1] - A.f_cmap_off[s];
    int* out = A.ea_tgt.data() + A.f_ea_off[s];
    for (int i = 0; i < len; i++)
      for (int j = 0; j <= i; j++) *out++ = m[i] * (m[i] + 1) / 2 + m[j];
  }
}

void dump_analysis(const Analysis& a_in, std::vector<int32_t>& out) {
  Analysis a = a_in;            // (test hook: the copy is fine)
  if (a.ea_tgt.empty()) expand_ea_tgt(a);
  if (a.el_tgt.empty()) expand_el_lists(a);
  a.el_src.assign((size_t)a.el_total, 0);   // (kept in the dump layout; only its length is used)
  out.clear();
  auto put = [&](int64_t v) { out.push_back((int32_t)v); };
  auto putv = [&](const std::vector<int>& v) { put((int64_t)v.size()); for (int x : v) out.push_back(x); };
  auto putv64 = [&](const std::vector<int64_t>& v) { put((int64_t)v.size()); for (int64_t x : v) out.push_back((int32_t)x); };
  put(a.n_nodes); put(a.n_scalars); put(a.n_fronts); put(a.n_levels); put(a.max_front);
  put(a.n_blocks); put(a.n_segs); put(a.L_size); put(a.U_size); put(a.H_size); put(a.J_size);
  putv(a.node_pos); putv(a.node_voff); putv(a.order);
  putv(a.f_p); putv(a.f_b); putv(a.f_poff); putv(a.f_parent); putv(a.f_level);
  putv64(a.f_Loff); putv64(a.f_Uoff);
  putv(a.f_bidx_off); putv(a.bidx); putv(a.f_child_off); putv(a.child); putv(a.f_cmap_off); putv(a.cmap);
  putv(a.level_off); putv(a.level_fronts);
  putv(a.f_asm_off); putv(a.asm_blk); putv(a.asm_lrow); putv(a.asm_lcol);
  putv(a.blk_rows); putv(a.blk_cols); putv(a.blk_size); putv(a.blk_nseg); putv64(a.blk_hoff);
  putv(a.seg_blk); putv(a.seg_c0); putv(a.seg_cnt); putv64(a.seg_hoff);
  putv(a.contrib);
  put(a.n_stages); put(a.n_groups); put(a.n_glevels);
  putv(a.stage_grp_off); putv(a.grp_lvl_off); putv(a.glvl_front_off); putv(a.glvl_fronts); putv(a.stage_max_front); putv(a.stage_max_width);
  putv(a.f_el_off); putv(a.el_src); putv(a.el_tgt); putv64(a.f_ea_off); putv(a.ea_tgt); putv(a.blk_doff); putv(a.blk_dst);
  putv(a.frec); putv(a.crec); putv(a.srec);
  putv(a.pidx); putv(a.obs_dir); putv(a.nd_segs);
}

}  // namespace pps
