// Host half of the ground-edge selection: see pps_edges_host.h.  Everything here is small sequential work on a few
// hundred segments per frame (the reference runs it in C++ and, through boost::python, in Python); the per-pixel
// stages before it are the kernels of pps_edges.hip.
#include "pps_edges_host.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <deque>
#include <limits>
#include <map>
#include <set>
#include <unordered_map>

namespace pps_edges_host {
namespace {

using Pt = std::pair<int, int>;   // (row, column)
using Line = std::array<float, 4>;
constexpr double kRefPi = 3.14159265;   // select_edge.cpp:18

inline uint32_t key_of(const Pt& p) { return ((uint32_t)(uint16_t)p.first << 16) | (uint16_t)p.second; }

struct Vec2 { float x, y; };
inline Vec2 operator-(Vec2 a, Vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline Vec2 operator+(Vec2 a, Vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline Vec2 operator*(float s, Vec2 a) { return {s * a.x, s * a.y}; }
inline float dot(Vec2 a, Vec2 b) { return a.x * b.x + a.y * b.y; }
inline float norm(Vec2 a) { return std::sqrt(a.x * a.x + a.y * a.y); }
inline Vec2 head(const Line& l) { return {l[0], l[1]}; }
inline Vec2 tail(const Line& l) { return {l[2], l[3]}; }

// matrix_utils.cpp:495-503
inline float fold_angle(float deg) { return deg > 90 ? deg - 180 : (deg < -90 ? deg + 180 : deg); }
// select_edge.cpp:111: atan2 in float, the scaling to degrees in double
inline float angle_deg(const Line& l) {
  const Vec2 v = tail(l) - head(l);
  return fold_angle((float)((double)std::atan2(v.y, v.x) / kRefPi * 180));
}
inline float angle_gap(float a, float b) { const float d = std::fabs(a - b); return std::min(d, 180 - d); }

struct DistProj { float dist, proj; };
// matrix_utils.cpp:331-351
DistProj dist_proj_to_line(Vec2 bg, Vec2 ed, Vec2 q) {
  const float len = norm(ed - bg);
  if (len < 0.001) return {norm(q - bg), -1.0f};
  float t = dot(q - bg, ed - bg) / len / len;
  const Vec2 foot = bg + t * (ed - bg);
  const float d = norm(q - foot);
  t = t > 1 ? 1 : t;
  t = t < 0 ? 0 : t;
  return {d, t};
}
// matrix_utils.cpp:318-329
float dist_to_line(Vec2 bg, Vec2 ed, Vec2 q) {
  const float len = norm(ed - bg);
  if (len < 0.001) return norm(q - bg);
  const float t = dot(q - bg, ed - bg) / len / len;
  return norm(q - (bg + t * (ed - bg)));
}
// matrix_utils.cpp:229-272 with nobottom = false
Vec2 ray_to_border(Vec2 pt, Vec2 dir, int w, int h) {
  auto inside = [](float v, int hi) { return 0 <= (int)v && (int)v <= hi; };
  if (dir.y < 0) {
    const float lam = (float)((0.0 - pt.y) / dir.y);
    if (lam >= 0) { const Vec2 hit = pt + lam * dir; if (inside(hit.x, w - 1)) return hit; }
  }
  if (dir.y > 0) {
    const float lam = (float)((h - 1.0 - pt.y) / dir.y);
    if (lam >= 0) { const Vec2 hit = pt + lam * dir; if (inside(hit.x, w - 1)) return hit; }
  }
  if (dir.x > 0) {
    const float lam = (float)((w - 1.0 - pt.x) / dir.x);
    if (lam >= 0) { const Vec2 hit = pt + lam * dir; if (inside(hit.y, h - 1)) return hit; }
  }
  if (dir.x < 0) {
    const float lam = (float)((0.0 - pt.x) / dir.x);
    if (lam >= 0) { const Vec2 hit = pt + lam * dir; if (inside(hit.y, h - 1)) return hit; }
  }
  return {-1, -1};
}

// the worst of ten samples along the line of the distance to the nearest contour point (select_edge.cpp:120-129)
float worst_contour_distance(const Line& l, const std::vector<float>& cxy) {
  float worst = -std::numeric_limits<float>::infinity();
  const size_t n = cxy.size() / 2;
  for (int k = 0; k < 10; k++) {
    const Vec2 s = head(l) + (float)(k / 10.0) * (tail(l) - head(l));
    float nearest = std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < n; i++) nearest = std::min(nearest, norm(Vec2{cxy[2 * i], cxy[2 * i + 1]} - s));
    worst = std::max(worst, nearest);
  }
  return worst;
}

// ---- intervals over x with the source line as payload (intervaltree semantics) ------------------------------------
struct Interval {
  double lo, hi;
  std::array<double, 4> data;
  bool operator<(const Interval& o) const { return std::tie(lo, hi, data) < std::tie(o.lo, o.hi, o.data); }
  bool operator==(const Interval& o) const { return lo == o.lo && hi == o.hi && data == o.data; }
};
using IntervalSet = std::set<Interval>;   // the tree is a set of (begin, end, data); iteration order = sorted()

}  // namespace

namespace {
// point -> contour id, open addressing (the keys are packed grid points; 0xffffffff never occurs: coordinates are non-negative int16)
struct PointMap {
  static constexpr uint32_t kEmpty = 0xffffffffu, kGone = 0xfffffffeu;
  std::vector<uint32_t> key; std::vector<int> val; uint32_t mask = 0;
  explicit PointMap(size_t n) { size_t c = 64; while (c < 4 * n) c <<= 1; key.assign(c, kEmpty); val.assign(c, -1); mask = (uint32_t)c - 1; }
  static uint32_t mix(uint32_t k) { k *= 0x9e3779b1u; return k ^ (k >> 15); }
  int find(uint32_t k) const {
    for (uint32_t h = mix(k) & mask;; h = (h + 1) & mask) { if (key[h] == k) return val[h]; if (key[h] == kEmpty) return -1; }
  }
  void set(uint32_t k, int v) {
    uint32_t slot = kEmpty;
    for (uint32_t h = mix(k) & mask;; h = (h + 1) & mask) {
      if (key[h] == k) { val[h] = v; return; }
      if (key[h] == kGone && slot == kEmpty) slot = h;
      if (key[h] == kEmpty) { if (slot == kEmpty) slot = h; break; }
    }
    key[slot] = k; val[slot] = v;
  }
  void erase(uint32_t k) {
    for (uint32_t h = mix(k) & mask;; h = (h + 1) & mask) { if (key[h] == k) { key[h] = kGone; return; } if (key[h] == kEmpty) return; }
  }
};
}  // namespace

Contour ground_contour(const CellSeg* segs, int n, float scale) {
  // skimage _assemble_contours: two dictionaries (first point -> contour, last point -> contour), contours numbered in order of
  // creation; a join keeps the older contour.  Round 6: the same steps on flat storage -- the points of all contours in one pool of
  // doubly linked nodes (a join relinks two lists instead of copying one), the dictionaries as open-addressing tables -- instead of a
  // std::map of deques and two std::unordered_map (861 cell segments of a VGA frame: 81 -> 14 us on the build machine).
  struct Node { Pt p; int next, prev; };
  struct Chain { int head = -1, tail = -1, count = 0; bool alive = false; };
  std::vector<Node> pool; pool.reserve(2 * (size_t)n + 2);
  std::vector<Chain> chains; chains.reserve((size_t)n / 2 + 2);
  PointMap starts((size_t)n + 1), ends((size_t)n + 1);
  auto node = [&](const Pt& p) { pool.push_back(Node{p, -1, -1}); return (int)pool.size() - 1; };
  auto push_back = [&](Chain& c, const Pt& p) { const int k = node(p); pool[k].prev = c.tail; pool[c.tail].next = k; c.tail = k; c.count++; };
  auto push_front = [&](Chain& c, const Pt& p) { const int k = node(p); pool[k].next = c.head; pool[c.head].prev = k; c.head = k; c.count++; };
  for (int i = 0; i < n; i++) {
    const Pt from{segs[i].fr, segs[i].fc}, to{segs[i].tr, segs[i].tc};
    if (from == to) continue;
    const int tail_id = starts.find(key_of(to));
    const int head_id = ends.find(key_of(from));
    if (tail_id >= 0 && head_id >= 0) {
      if (tail_id == head_id) {
        push_back(chains[head_id], to);
        starts.erase(key_of(to)); ends.erase(key_of(from));
      } else if (tail_id > head_id) {                               // the tail contour is appended to the (older) head contour
        Chain& hd = chains[head_id]; Chain& tl = chains[tail_id];
        const Pt tl_last = pool[tl.tail].p;
        pool[hd.tail].next = tl.head; pool[tl.head].prev = hd.tail; hd.tail = tl.tail; hd.count += tl.count;
        starts.erase(key_of(to)); ends.erase(key_of(tl_last)); tl = Chain();
        ends.erase(key_of(from));
        ends.set(key_of(pool[hd.tail].p), head_id);
      } else {                                                      // the head contour goes in front of the (older) tail contour
        Chain& hd = chains[head_id]; Chain& tl = chains[tail_id];
        const Pt hd_first = pool[hd.head].p;
        pool[hd.tail].next = tl.head; pool[tl.head].prev = hd.tail; tl.head = hd.head; tl.count += hd.count;
        starts.erase(key_of(hd_first)); ends.erase(key_of(from)); hd = Chain();
        starts.erase(key_of(to));
        starts.set(key_of(pool[tl.head].p), tail_id);
      }
    } else if (tail_id < 0 && head_id < 0) {
      const int id = (int)chains.size();
      chains.emplace_back();
      Chain& c = chains.back();
      c.alive = true; c.head = c.tail = node(from); c.count = 1;
      push_back(c, to);
      starts.set(key_of(from), id); ends.set(key_of(to), id);
    } else if (tail_id >= 0) {
      push_front(chains[tail_id], from);
      starts.erase(key_of(to)); starts.set(key_of(from), tail_id);
    } else {
      push_back(chains[head_id], to);
      ends.erase(key_of(from)); ends.set(key_of(to), head_id);
    }
  }
  Contour out;
  const Chain* pick = nullptr;
  double longest = -1.0;
  for (const Chain& c : chains) {                                   // (creation order = the iteration order of the dictionary of contours)
    if (!c.alive) continue;
    out.n_contours++;
    const Pt a = pool[c.head].p, b = pool[c.tail].p;
    const double dr = (double)a.first - b.first, dc = (double)a.second - b.second;
    const double gap = std::sqrt(dr * dr + dc * dc);
    if (gap > longest) { longest = gap; pick = &c; }
  }
  if (!pick) return out;
  out.n_points = pick->count;
  int k = pick->head;
  for (int i = 0; i + 1 < pick->count; i++, k = pool[k].next) {
    if (i % 20) continue;
    out.xy.push_back((float)pool[k].p.second * scale);
    out.xy.push_back((float)pool[k].p.first * scale);
  }
  return out;
}

std::vector<float> interval_tree_optimization(const std::vector<float>& lines, double overlap_thre) {
  const int n = (int)lines.size() / 4;
  std::vector<float> result;
  if (n == 0) return result;
  auto line_of = [&](int i) { return Line{lines[4 * i], lines[4 * i + 1], lines[4 * i + 2], lines[4 * i + 3]}; };
  auto as_interval = [&](int i) {
    const Line l = line_of(i);
    return Interval{l[0], l[2], {l[0], l[1], l[2], l[3]}};
  };
  std::vector<float> length(n);
  for (int i = 0; i < n; i++) length[i] = norm(tail(line_of(i)) - head(line_of(i)));
  std::vector<int> todo(n);
  for (int i = 0; i < n; i++) todo[i] = i;
  IntervalSet cover;
  auto take_longest_of = [&](const std::vector<int>& candidates) {   // positions in todo; first maximum wins
    int best = candidates[0];
    for (int c : candidates) if (length[todo[c]] > length[todo[best]]) best = c;
    cover.insert(as_interval(todo[best]));
    todo.erase(todo.begin() + best);
  };
  { std::vector<int> all(n); for (int i = 0; i < n; i++) all[i] = i; take_longest_of(all); }
  while (!todo.empty()) {
    std::vector<int> fits;
    for (int c = 0; c < (int)todo.size(); c++) {
      const double q0 = lines[4 * todo[c]], q1 = lines[4 * todo[c] + 2];
      double shared = 0;
      if (q0 < q1)
        for (const Interval& iv : cover)
          if (iv.lo < q1 && iv.hi > q0) shared += std::min(std::min(q1 - iv.lo, iv.hi - q0), std::min(std::fabs(q1 - q0), iv.hi - iv.lo));
      if (shared < overlap_thre) fits.push_back(c);
    }
    if (fits.empty()) break;
    take_longest_of(fits);
  }
  // split_overlaps(): every interval is cut at every boundary inside it
  const size_t before = cover.size();
  {
    std::set<double> bounds;
    for (const Interval& iv : cover) { bounds.insert(iv.lo); bounds.insert(iv.hi); }
    if (bounds.size() > 2) {
      IntervalSet pieces;
      for (auto lo = bounds.begin(), hi = std::next(lo); hi != bounds.end(); ++lo, ++hi)
        for (const Interval& iv : cover)
          if (iv.lo <= *lo && *lo < iv.hi) pieces.insert(Interval{*lo, *hi, iv.data});
      cover.swap(pieces);
    }
  }
  const bool overlapped = cover.size() != before;
  if (overlapped) {
    // pieces over the same x range: the one whose source line is shorter in x goes
    std::vector<Interval> sorted(cover.begin(), cover.end());
    IntervalSet drop;
    for (size_t i = 0; i < sorted.size(); i++)
      for (size_t j = i + 1; j < sorted.size(); j++)
        if (sorted[i].lo == sorted[j].lo && sorted[i].hi == sorted[j].hi)
          drop.insert((sorted[i].data[2] - sorted[i].data[0]) < (sorted[j].data[2] - sorted[j].data[0]) ? sorted[i] : sorted[j]);
    for (const Interval& iv : drop) cover.erase(iv);
    // neighbouring pieces of one source line are joined again
    for (int round = 0; round < 100; round++) {
      sorted.assign(cover.begin(), cover.end());
      bool joined = false;
      for (size_t i = 0; i < sorted.size() && !joined; i++)
        for (size_t j = i + 1; j < sorted.size(); j++)
          if ((sorted[i].lo == sorted[j].hi || sorted[i].hi == sorted[j].lo) && sorted[i].data == sorted[j].data) {
            const Interval m{std::min(sorted[i].lo, sorted[j].lo), std::max(sorted[i].hi, sorted[j].hi), sorted[j].data};
            cover.erase(sorted[i]); cover.erase(sorted[j]); cover.insert(m);
            joined = true;
            break;
          }
      if (!joined) break;
    }
  }
  for (const Interval& iv : cover) {
    const auto& raw = iv.data;
    if (overlapped && (iv.lo != raw[0] || iv.hi != raw[2])) {
      const double f_lo = (iv.lo - raw[0]) / (raw[2] - raw[0]), f_hi = (iv.hi - raw[0]) / (raw[2] - raw[0]);
      result.insert(result.end(), {(float)iv.lo, (float)(int)(f_lo * (raw[3] - raw[1]) + raw[1]), (float)iv.hi,
                                   (float)(int)(f_hi * (raw[3] - raw[1]) + raw[1])});
    } else
      result.insert(result.end(), {(float)raw[0], (float)raw[1], (float)raw[2], (float)raw[3]});
  }
  return result;
}

Selection select(const std::vector<float>& cxy, int width, int height, const float* lsd, int n_lsd, const pps_edge_params& prm) {
  Selection out;
  std::vector<Line> lines;
  // steps 1 and 2: length, image border, near-vertical, distance to the CNN boundary; then left end first
  for (int i = 0; i < n_lsd; i++) {
    const Line l{lsd[4 * i], lsd[4 * i + 1], lsd[4 * i + 2], lsd[4 * i + 3]};
    if (norm(tail(l) - head(l)) < prm.pre_minium_len) continue;
    const double m = prm.pre_boundary_thre;
    const bool hugs_border = (l[0] < m && l[2] < m) || (l[0] > width - m && l[2] > width - m) || (l[1] < m && l[3] < m) ||
                             (l[1] > height - m && l[3] > height - m);
    if (hugs_border) continue;
    if (!(std::fabs(std::fabs(angle_deg(l)) - 90) > prm.pre_vertical_thre)) continue;
    if (!(worst_contour_distance(l, cxy) < prm.pre_contour_close_thre)) continue;
    lines.push_back(l[0] > l[2] ? Line{l[2], l[3], l[0], l[1]} : l);
  }
  // step 3: end-to-start chaining of near-collinear lines, restarting after every change
  for (int round = 0; round < 100; round++) {
    bool changed = false;
    std::vector<float> ang(lines.size());
    for (size_t i = 0; i < lines.size(); i++) ang[i] = angle_deg(lines[i]);
    for (size_t a = 0; a < lines.size() && !changed; a++)
      for (size_t b = a + 1; b < lines.size(); b++) {
        if (!(angle_gap(ang[a], ang[b]) < prm.pre_merge_angle_thre)) continue;
        const float a_end_to_b = norm(tail(lines[a]) - head(lines[b])), b_end_to_a = norm(tail(lines[b]) - head(lines[a]));
        if (a_end_to_b < prm.pre_merge_dist_thre) { lines[a][2] = lines[b][2]; lines[a][3] = lines[b][3]; }
        else if (b_end_to_a < prm.pre_merge_dist_thre) { lines[a][0] = lines[b][0]; lines[a][1] = lines[b][1]; }
        else continue;
        lines.erase(lines.begin() + b);
        changed = true;
        break;
      }
    if (!changed) break;
  }
  // step 4: two near-parallel lines that project onto each other -> one of them goes
  for (int round = 0; round < 100; round++) {
    bool changed = false;
    std::vector<float> ang(lines.size());
    for (size_t i = 0; i < lines.size(); i++) ang[i] = angle_deg(lines[i]);
    for (size_t a = 0; a < lines.size() && !changed; a++)
      for (size_t b = a + 1; b < lines.size(); b++) {
        if (!(angle_gap(ang[a], ang[b]) < prm.pre_proj_angle_thre)) continue;
        const Line &A = lines[a], &B = lines[b];
        const DistProj a0 = dist_proj_to_line(head(B), tail(B), head(A)), a1 = dist_proj_to_line(head(B), tail(B), tail(A));
        const DistProj b0 = dist_proj_to_line(head(A), tail(A), head(B)), b1 = dist_proj_to_line(head(A), tail(A), tail(B));
        const double far = prm.pre_proj_dist_thre;
        if (!(a0.dist < far && a1.dist < far && b0.dist < far && b1.dist < far)) continue;
        const float a_on_b = std::fabs(a0.proj - a1.proj), b_on_a = std::fabs(b0.proj - b1.proj);
        if (!(a_on_b > prm.pre_proj_cover_thre || b_on_a > prm.pre_proj_cover_thre)) continue;
        size_t victim;
        if (std::min(a_on_b, b_on_a) < prm.pre_proj_cover_large_thre) victim = a_on_b > b_on_a ? b : a;
        else victim = worst_contour_distance(A, cxy) > worst_contour_distance(B, cxy) ? a : b;
        lines.erase(lines.begin() + victim);
        changed = true;
        break;
      }
    if (!changed) break;
  }
  if (lines.empty()) return out;
  // step 5
  std::vector<float> flat;
  for (const Line& l : lines) flat.insert(flat.end(), l.begin(), l.end());
  const std::vector<float> opt = interval_tree_optimization(flat, prm.interval_overlap_thre);
  // step 6
  std::vector<Line> segs;
  for (size_t i = 0; i + 3 < opt.size(); i += 4) {
    const Line l{opt[i], opt[i + 1], opt[i + 2], opt[i + 3]};
    if (norm(tail(l) - head(l)) > prm.post_short_thre) segs.push_back(l);
  }
  if (segs.empty()) return out;
  for (int round = 0; round < 100; round++) {   // bind near end points of successive pieces to their integer mid-point
    bool changed = false;
    for (size_t s = 0; s + 1 < segs.size(); s++) {
      Line &A = segs[s], &B = segs[s + 1];
      if ((A[2] != B[0] || A[3] != B[1]) && norm(tail(A) - head(B)) < prm.post_bind_dist_thre) {
        const int mx = (int)((A[2] + B[0]) / 2), my = (int)((A[3] + B[1]) / 2);
        A[2] = B[0] = (float)mx; A[3] = B[1] = (float)my;
        changed = true;
      }
    }
    if (!changed) break;
  }
  for (int round = 0; round < 100; round++) {   // successive near-collinear pieces become one
    bool changed = false;
    for (size_t s = 0; s + 1 < segs.size(); s++) {
      const Line &A = segs[s], &B = segs[s + 1];
      if (!(angle_gap(angle_deg(A), angle_deg(B)) < prm.post_merge_angle_thre)) continue;
      const double near = prm.post_merge_dist_thre;
      const bool a_on_b = (dist_to_line(head(B), tail(B), head(A)) < near) && (dist_to_line(head(B), tail(B), tail(A)) < near);
      const bool b_on_a = (dist_to_line(head(A), tail(A), head(B)) < near) && (dist_to_line(head(A), tail(A), tail(B)) < near);
      if (a_on_b || b_on_a) {
        segs[s][2] = B[2]; segs[s][3] = B[3];
        segs.erase(segs.begin() + s + 1);
        changed = true;
        break;
      }
    }
    if (!changed) break;
  }
  {   // first / last piece run on to the image border if that stays close to the CNN boundary
    Line& first = segs.front(); Line& last = segs.back();
    const Vec2 start0 = head(first), end0 = tail(last);
    const Vec2 s_hit = ray_to_border(tail(first), head(first) - tail(first), width, height);
    const Vec2 e_hit = ray_to_border(head(last), tail(last) - head(last), width, height);
    first[0] = (float)(int)s_hit.x; first[1] = (float)(int)s_hit.y;
    last[2] = (float)(int)e_hit.x; last[3] = (float)(int)e_hit.y;
    const float d_first = worst_contour_distance(first, cxy), d_last = worst_contour_distance(last, cxy);
    if (d_first > prm.post_extend_thre) { first[0] = start0.x; first[1] = start0.y; }
    if (d_last > prm.post_extend_thre) { last[2] = end0.x; last[3] = end0.y; }
  }
  for (size_t s = 0; s < segs.size(); s++) {
    out.open_segs.insert(out.open_segs.end(), segs[s].begin(), segs[s].end());
    if (s > 0 && (segs[s - 1][2] != segs[s][0] || segs[s - 1][3] != segs[s][1]))
      out.closed_segs.insert(out.closed_segs.end(), {segs[s - 1][2], segs[s - 1][3], segs[s][0], segs[s][1]});
    out.open_in_closed.push_back((float)(out.closed_segs.size() / 4));
    out.closed_segs.insert(out.closed_segs.end(), segs[s].begin(), segs[s].end());
  }
  return out;
}

}  // namespace pps_edges_host
