// pps_upload.cpp -- from the host tables to HBM: compaction + symbolic analysis (once per topology; incremental for frame
// loops), the two device arenas, the difference upload (pinned mirror, patch buffer + scatter kernel), state and measurement
// transfers.  pps_graph.h lists the other implementation files.
#include "pps_graph.h"

using namespace pps;
using namespace pps_impl;

namespace pps_impl {

void free_device(pps_graph* g) {
  for (void* p : g->allocs) (void)hipFree(p);
  g->allocs.clear();
  // arenas are kept; grow them when the last layout spilled into fallback allocations
  for (pps_graph::Arena* a : {&g->up, &g->scr}) {
    const size_t want = (a == &g->up ? std::max(a->off, g->up_high) : a->off) + a->spill;
    if (a->spill > 0 || a->base == nullptr) {
      if (a->base) (void)hipFree(a->base);
      a->cap = std::max<size_t>(size_t(1) << 20, 2 * want);
      if (hipMalloc(reinterpret_cast<void**>(&a->base), a->cap) != hipSuccess) { a->base = nullptr; a->cap = 0; }
      if (a == &g->up) { g->up_slots.clear(); g->up_high = 0; g->up_unknown = true; }
    }
    a->off = 0; a->spill = 0;
  }
  g->up_cursor = 0; g->up_patches.clear();
  if (g->stage_cap < g->up.cap) {
    if (g->stage) (void)hipHostFree(g->stage);
    g->stage = nullptr; g->stage_cap = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&g->stage), g->up.cap, hipHostMallocDefault) == hipSuccess) g->stage_cap = g->up.cap;
    g->up_unknown = true;
  }
  g->stage_lo = g->stage_hi = 0;
  g->dev = DevGraph();
}

void release_arenas(pps_graph* g) {
  for (pps_graph::Arena* a : {&g->up, &g->scr}) { if (a->base) (void)hipFree(a->base); a->base = nullptr; a->cap = a->off = a->spill = 0; }
  if (g->stage) (void)hipHostFree(g->stage);
  g->stage = nullptr; g->stage_cap = 0;
  if (g->patch_host) (void)hipHostFree(g->patch_host);
  if (g->state_pin) (void)hipHostFree(g->state_pin);
  g->state_pin = nullptr; g->state_pin_cap = 0; g->pin_holds_est = false;
  g->patch_host = nullptr; g->patch_cap = 0;
  g->up_slots.clear(); g->up_high = 0; g->up_unknown = true;
}

// bytes [0, n) of `src` against the mirror at offset o: record (and copy into the mirror) the range that differs
// same_prefix: leading bytes the caller knows to be in the mirror already (pps_graph::an_seq): the comparison starts behind them
void up_diff(pps_graph* g, size_t o, const char* src, size_t n, bool force, size_t same_prefix) {
  if (n == 0) return;
  char* mir = g->stage + o;
  g->up_bytes_total += n;
  if (g->up_unknown || force) { memcpy(mir, src, n); g->up_patches.push_back(pps_graph::UpPatch{o, n, false}); return; }
  same_prefix = std::min(same_prefix, n) & ~size_t(63);
  if (same_prefix && g->verify_hints && memcmp(mir, src, same_prefix) != 0) g->hint_violation = true;
  // first and last 64-byte chunk that differs from what the device holds (4 KB strides first: most arrays of a frame loop
  // are unchanged from end to end, or up to a short tail)
  size_t lo = same_prefix, hi = n;
  while (lo + 4096 <= hi && memcmp(mir + lo, src + lo, 4096) == 0) lo += 4096;
  while (lo + 64 <= hi && memcmp(mir + lo, src + lo, 64) == 0) lo += 64;
  if (lo + 64 > hi && memcmp(mir + lo, src + lo, hi - lo) == 0) return;      // identical
  while (hi >= lo + 4096 && memcmp(mir + hi - 4096, src + hi - 4096, 4096) == 0) hi -= 4096;
  while (hi >= lo + 64 && memcmp(mir + hi - 64, src + hi - 64, 64) == 0) hi -= 64;
  lo &= ~size_t(15);
  memcpy(mir + lo, src + lo, hi - lo);
  g->up_patches.push_back(pps_graph::UpPatch{o + lo, hi - lo, false});
}

// The k-th upload of a layout goes where the k-th upload of the previous layout went, as long as it fits the slot.
// rows > 0: an SoA array of `rows` rows with leading dimension ld of which the first `used` entries per row are live -- the
// rows are compared one by one (appending a factor touches the end of every row, not the array from end to end).

// send what differs: everything in one copy when the device content is unknown or most of it changed, else the patches
int flush_uploads(pps_graph* g) {
  // callers: upload_all (after its opening stream sync) and the frame tables of pps_refresh_measurements (which settles
  // up_inflight first) -- the pinned mirror and the patch buffer are never rewritten under a copy that still reads them
  size_t sent = 0;
  for (const auto& pt : g->up_patches) sent += pt.len;
  g->up_bytes_sent = sent;
  if (g->up_patches.empty()) { g->up_bytes_total = 0; return PPS_OK; }
  // One copy of the whole arena only when the device content is unknown.  Otherwise nothing but the changed pieces may be
  // written: the span between two pieces can hold what kernels have refreshed behind the mirror's back (the observation
  // measurements that stay on the device) -- a copy "from the first to the last change" would put stale values over them.
  if (g->up_unknown) {
    size_t lo = 0, hi = 0;
    for (const auto& pt : g->up_patches) hi = std::max(hi, pt.off + pt.len);
    hi = std::max(hi, g->up_high); hi = std::min(hi, g->stage_cap);
    HIP_TRY(g, hipMemcpyAsync(g->up.base + lo, g->stage + lo, hi - lo, hipMemcpyHostToDevice, g->stream));
    g->up_bytes_sent = hi - lo;
  } else {
    // A table of (destination, source, bytes, unit) records in pinned host memory; k_scatter_patches reads the table AND the pieces -- straight
    // from the pinned mirror, whose offsets are the arena's -- over the bus (round 6: no patch buffer, no host copy of the pieces into it, no
    // copy launch in front of the kernel: 1.7 copies of 63 KB per frame of the frame loop).  Piece lengths are rounded up to 16 bytes except
    // where the bytes behind a piece may be newer on the device than in the mirror (exact8).
    const size_t np = g->up_patches.size();
    const size_t need = np * 32;
    if (need > g->patch_cap) {
      if (g->patch_host) (void)hipHostFree(g->patch_host);
      g->patch_host = nullptr; g->patch_cap = 0;
      const size_t cap = std::max<size_t>(1 << 12, 2 * need);
      HIP_TRY(g, hipHostMalloc(reinterpret_cast<void**>(&g->patch_host), cap, hipHostMallocDefault));
      g->patch_cap = cap;
    }
    long long* tab = reinterpret_cast<long long*>(g->patch_host);
    for (size_t i = 0; i < np; i++) {
      const auto& pt = g->up_patches[i];
      const size_t len = pt.exact8 ? pt.len : ((pt.len + 15) & ~size_t(15));      // (exact pieces are multiples of 8; the arena's capacity is a multiple of 16)
      tab[4 * i + 0] = (long long)pt.off; tab[4 * i + 1] = (long long)pt.off; tab[4 * i + 2] = (long long)len; tab[4 * i + 3] = pt.exact8 ? 1 : 0;
    }
    HIP_TRY(g, launch_scatter_patches(g->patch_host, g->stage, (int)np, g->up.base, g->stream));
    // (table and mirror are written again by the next upload_all / refresh, which wait for the stream first: up_inflight)
  }
  g->up_unknown = false;
  g->up_patches.clear();
  g->stage_lo = g->stage_hi = 0;
  return PPS_OK;
}

// PPS_DEBUG_VERIFY_UPLOAD=1: after a flush, the arena on the device must equal the pinned mirror (except the observation
// measurements, which kernels refresh behind the mirror's back) -- PPS_ESTATE if not.  tests/test_gpu_pipeline.py runs a frame
// loop under it.
int verify_uploads(pps_graph* g, const char* where) {
  if (g->hint_violation) { g->hint_violation = false; return fail(g, PPS_ESTATE, std::string("upload verification (") + where + "): an array differs from the mirror inside the part the analysis reported as kept"); }
  if (!g->sw.verify_upload || g->up_high == 0 || g->up.spill) return PPS_OK;
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  std::vector<char> dev(g->up_high);
  HIP_TRY(g, hipMemcpy(dev.data(), g->up.base, g->up_high, hipMemcpyDeviceToHost));
  for (size_t k = 0; k < g->up_slots.size() && k < g->up_cursor; k++) {
    if (k == g->slot_obs_meas || k == g->slot_lp_meas) continue;
    const size_t o = g->up_slots[k].off, n = std::min(g->up_slots[k].cap, g->up_high - std::min(g->up_high, o));
    if (o >= g->up_high) continue;
    if (memcmp(dev.data() + o, g->stage + o, n) != 0) {
      size_t b = 0; while (b < n && dev[o + b] == g->stage[o + b]) b++;
      return fail(g, PPS_ESTATE, std::string("upload verification (") + where + "): slot " + std::to_string(k) + " (offset " + std::to_string(o) + ", capacity " +
                                     std::to_string(g->up_slots[k].cap) + ") differs from the mirror at byte " + std::to_string(b));
    }
  }
  return PPS_OK;
}

int ensure_device(pps_graph* g) {
  if (g->dev_ready) return PPS_OK;
  HIP_TRY(g, hipSetDevice(g->props.device));
  HIP_TRY(g, hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
  HIP_TRY(g, hipHostMalloc(reinterpret_cast<void**>(&g->host_result), 12 * sizeof(double), hipHostMallocDefault));
  HIP_TRY(g, hipEventCreate(&g->ev[0]));
  HIP_TRY(g, hipEventCreate(&g->ev[1]));
  g->dev_ready = true;
  return PPS_OK;
}

// Offsets of the four per-type slabs of the J buffer: [plane obs | odometry | pose priors | plane priors].  Every slab is
// sized for a capacity that grows in powers of two, so that a graph that gains a few factors per frame keeps all of its J
// offsets -- and with them the whole contribution list of the H-block kernel -- from one frame to the next.
int64_t j_capacity(int64_t n) { int64_t c = 16; while (c < n) c <<= 1; return c; }
void j_bases(const pps_graph* g, int64_t base[4], int64_t* total) {
  const int64_t n_pp = g->fslot_ids[F_POSE_PRIOR].size(), n_odo = g->fslot_ids[F_ODOMETRY].size(),
                n_obs = g->fslot_ids[F_PLANE_OBS].size(), n_lp = g->fslot_ids[F_PLANE_PRIOR].size();
  const int64_t joff_obs = 0, joff_odo = j_capacity(n_obs) * 30, joff_pp = joff_odo + j_capacity(n_odo) * 78,
                joff_lp = joff_pp + j_capacity(n_pp) * 42;
  base[F_POSE_PRIOR] = joff_pp; base[F_ODOMETRY] = joff_odo; base[F_PLANE_OBS] = joff_obs; base[F_PLANE_PRIOR] = joff_lp;
  if (total) *total = joff_lp + j_capacity(n_lp) * 12;
}
// the product records (pps_symbolic.h: kPSize) in slabs of the same capacities: a product offset moves exactly when the J offset does
void p_bases(const pps_graph* g, int64_t base[4], int64_t* total) {
  const int64_t n_pp = g->fslot_ids[F_POSE_PRIOR].size(), n_odo = g->fslot_ids[F_ODOMETRY].size(),
                n_obs = g->fslot_ids[F_PLANE_OBS].size(), n_lp = g->fslot_ids[F_PLANE_PRIOR].size();
  const int64_t p_obs = 0, p_odo = j_capacity(n_obs) * kPSize[F_PLANE_OBS], p_pp = p_odo + j_capacity(n_odo) * kPSize[F_ODOMETRY],
                p_lp = p_pp + j_capacity(n_pp) * kPSize[F_POSE_PRIOR];
  base[F_POSE_PRIOR] = p_pp; base[F_ODOMETRY] = p_odo; base[F_PLANE_OBS] = p_obs; base[F_PLANE_PRIOR] = p_lp;
  if (total) *total = p_lp + j_capacity(n_lp) * kPSize[F_PLANE_PRIOR];
}

// ---- compaction + symbolic analysis (host only) -------------------------------------------
static int run_analysis_impl(pps_graph* g);
// A failed analysis may have rewritten g->an in place (the incremental path) before it gave up: the handle is marked as not
// analysed, so that no later upload sends a half-rewritten Analysis to the device -- the next solve runs the analysis again.
int run_analysis(pps_graph* g) {
  const int rc = run_analysis_impl(g);
  if (rc != PPS_OK) { g->analyzed = false; g->analysis_stale = true; g->topo_dirty = true; g->cmp_valid = false; g->an_seq++; g->an_base_seq = -1; }
  return rc;
}
static int run_analysis_impl(pps_graph* g) {
  const double t0 = now_s();
  std::vector<SymNode>& sn = g->sym_nodes;
  std::vector<SymFactor>& sf = g->sym_factors;
  // A graph that only grew since the last analysis (the frame loop) appends to the compacted tables instead of walking
  // every node and factor again; re-popping edges are ordered behind the fixed ones, which moves slots: they take the full path.
  bool append = g->cmp_valid && g->grown_only && g->n_analyses > 0 && !g->cmp_has_repop && g->cmp_nodes <= g->nodes.size() &&
                g->cmp_factors <= g->factors.size() && !g->sw.no_incr_compact;
  for (size_t i = g->cmp_factors; append && i < g->factors.size(); i++)
    append = !g->factors[i].deleted && !(g->factors[i].type == F_PLANE_OBS && g->factors[i].repop);
  for (size_t i = g->cmp_nodes; append && i < g->nodes.size(); i++) append = !g->nodes[i].deleted;
  if (append) {
    for (size_t i = g->cmp_nodes; i < g->nodes.size(); i++) {
      HostNode& n = g->nodes[i];
      n.compact = (int)sn.size();
      if (n.type == NODE_POSE) { n.slot = (int)g->pose_ids.size(); g->pose_ids.push_back((int)i); sn.push_back({NODE_POSE, 6, n.slot}); }
      else { n.slot = (int)g->plane_ids.size(); g->plane_ids.push_back((int)i); sn.push_back({NODE_PLANE, 3, -1}); }
    }
    for (size_t i = g->cmp_factors; i < g->factors.size(); i++) {
      HostFactor& f = g->factors[i];
      f.slot = (int)g->fslot_ids[f.type].size();
      g->fslot_ids[f.type].push_back((int)i);
    }
    g->n_obs_fixed = (int)g->fslot_ids[F_PLANE_OBS].size();
    int64_t base[4], j_total = 0, pbase[4], p_total = 0;
    j_bases(g, base, &j_total);
    p_bases(g, pbase, &p_total);
    if (j_total > 0x7fffffffLL || p_total > 0x7fffffffLL) return fail(g, PPS_ENOMEM, "graph too large for int32 J offsets");
    if (memcmp(base, g->cmp_base, sizeof(base)) != 0) {          // a J slab outgrew its capacity: every offset moves
      int cnt[4] = {0, 0, 0, 0};
      for (SymFactor& q : sf) { const int k = cnt[q.type]++; q.joff = (int)(base[q.type] + (int64_t)k * kJSize[q.type]); q.poff = (int)(pbase[q.type] + (int64_t)k * kPSize[q.type]); }
      memcpy(g->cmp_base, base, sizeof(base));
    }
    for (size_t i = g->cmp_factors; i < g->factors.size(); i++) {
      const HostFactor& f = g->factors[i];
      SymFactor q;
      q.type = f.type;
      q.a = g->nodes[f.a].compact;
      q.b = f.b >= 0 ? g->nodes[f.b].compact : -1;
      q.joff = (int)(base[f.type] + (int64_t)f.slot * kJSize[f.type]);
      q.poff = (int)(pbase[f.type] + (int64_t)f.slot * kPSize[f.type]);
      q.direct_ok = f.type == F_PLANE_OBS ? 1 : 0;
      sf.push_back(q);
    }
  } else {
  g->pose_ids.clear(); g->plane_ids.clear();
  for (int t = 0; t < 4; t++) g->fslot_ids[t].clear();
  sn.clear(); sf.clear();
  for (size_t i = 0; i < g->nodes.size(); i++) {
    HostNode& n = g->nodes[i];
    if (n.deleted) { n.compact = n.slot = -1; continue; }
    n.compact = (int)sn.size();
    if (n.type == NODE_POSE) { n.slot = (int)g->pose_ids.size(); g->pose_ids.push_back((int)i); sn.push_back({NODE_POSE, 6, n.slot}); }
    else { n.slot = (int)g->plane_ids.size(); g->plane_ids.push_back((int)i); sn.push_back({NODE_PLANE, 3, -1}); }
  }
  // plane observations with a fixed measurement first, the re-popping ones (Factor2) behind them
  g->cmp_has_repop = false;
  for (int pass = 0; pass < 2; pass++)
    for (size_t i = 0; i < g->factors.size(); i++) {
      HostFactor& f = g->factors[i];
      if (f.deleted) { f.slot = -1; continue; }
      if ((f.type == F_PLANE_OBS && f.repop) != (pass == 1)) continue;
      if (pass == 1) g->cmp_has_repop = true;
      f.slot = (int)g->fslot_ids[f.type].size();
      g->fslot_ids[f.type].push_back((int)i);
    }
  g->n_obs_fixed = 0;
  for (int id : g->fslot_ids[F_PLANE_OBS]) g->n_obs_fixed += g->factors[id].repop ? 0 : 1;
  int64_t base[4], j_total = 0, pbase[4], p_total = 0;
  j_bases(g, base, &j_total);
  p_bases(g, pbase, &p_total);
  if (j_total > 0x7fffffffLL || p_total > 0x7fffffffLL) return fail(g, PPS_ENOMEM, "graph too large for int32 J offsets");
  memcpy(g->cmp_base, base, sizeof(base));
  sf.reserve(g->factors.size());
  bool any_deleted = false;
  for (size_t i = 0; i < g->factors.size(); i++) {
    const HostFactor& f = g->factors[i];
    if (f.deleted) { any_deleted = true; continue; }
    SymFactor s;
    s.type = f.type;
    s.a = g->nodes[f.a].compact;
    s.b = f.b >= 0 ? g->nodes[f.b].compact : -1;
    s.joff = (int)(base[f.type] + (int64_t)f.slot * kJSize[f.type]);
    s.poff = (int)(pbase[f.type] + (int64_t)f.slot * kPSize[f.type]);
    s.direct_ok = (f.type == F_PLANE_OBS && !f.repop) ? 1 : 0;
    sf.push_back(s);
  }
  // the append path relies on: table index == host index order with nothing skipped
  g->cmp_valid = !any_deleted && sn.size() == g->nodes.size();
  }
  g->cmp_nodes = g->nodes.size(); g->cmp_factors = g->factors.size();
  // band depth: 3 levels per launch when the solve is latency bound, 2 when the lower levels are throughput bound (C3: 5 360 fronts;
  // 637 vs 710 us).  (Up to round 4: 4 levels, 113.3 vs 115.0 us per C2 LM iteration with 3.  Round 5: a group of 4 + 2 + 1 fronts on
  // eight waves has a wave for every front, so every separator front is assembled ahead of its level while the leaves are eliminated,
  // and the top group -- levels 6 .. 8 of C2's nine -- is factored and solved by one launch with the data-flow back-substitution:
  // 60.9 against 63.7 us per C2 LM iteration with 4, same box.)
  g->aprm.band_levels = 0;        // chosen by the analysis from the tree: three levels per band below 2 400 poses, four above -- two where a front needs the fifteen-tile kernel (pps_symbolic.cpp)
  // H-block segments (contributions reduced by one wave of K2): short on small graphs, where the few long segments
  // (ground plane, 32 contributions = 16 dependent load rounds) are K2's critical path; long on large ones, where the
  // number of waves is (C2: 23.3 -> 18.7 us with 8; C3: 82 -> 102 us)
  // (round 4: a plane observation reaches K2 as ONE coalesced load per lane -- its product record -- so a wave sums up to 64
  // contributions per segment: every landmark but the ground plane is one segment, and no second reduction pass)
  g->aprm.seg_len = 64;
  // a graph that is re-analysed after pure appends is a frame loop: absolute cut positions keep the left part of its tree
  g->aprm.aligned_cuts = (g->n_analyses > 0 && g->grown_only) ? 1 : 0;
  // ... whose groups of 4 leaves (3 levels per launch) keep every front of a level on its own wave (C5: 1 580 vs 1 500 frames/s; up to round 5
  // its aligned cuts also left fronts of 65 .. 80 rows: AnalysisParams::front_rows / cut_shift now keep every front of the loop within 63)
  if (g->aprm.aligned_cuts && g->pose_ids.size() < 4000) g->aprm.band_levels = 3;
  g->aprm.band_rows = band_front_limit();
  const char* msg = "";
  try {
  if (g->sw.analysis_timing) fprintf(stderr, "[analysis] %-22s %8.3f ms\n", "compaction (api)", 1e3 * (now_s() - t0));
  if (!g->acache) g->acache = analysis_cache_new();
  if (!analyze(sn, sf, g->aprm, g->an, &msg, g->sw.no_incremental ? nullptr : g->acache))
    return fail(g, PPS_EINVAL, std::string("analysis failed: ") + msg);
  // fronts beyond the wave-per-front kernels (loop-closure separators) run in the dense-front form, whose cost is
  // per tree level: split their supernodes into 64-pivot chunks instead of 48 (a quarter fewer levels)
  if (g->an.max_front > band_front_limit() && g->aprm.max_pivots < dense_front_max_pivots()) {
    AnalysisParams wide = g->aprm;
    wide.max_pivots = dense_front_max_pivots();
    if (!analyze(sn, sf, wide, g->an, &msg, g->sw.no_incremental ? nullptr : g->acache))
      return fail(g, PPS_EINVAL, std::string("analysis failed: ") + msg);
  }
  } catch (const std::bad_alloc&) {
    g->an = Analysis();
    return fail(g, PPS_ENOMEM, "symbolic analysis ran out of host memory (fronts too wide for this ordering)");
  }
  g->level_max_front.assign(g->an.n_levels, 0);
  for (int s = 0; s < g->an.n_fronts; s++) {
    int& m = g->level_max_front[g->an.f_level[s]];
    m = std::max(m, g->an.f_p[s] + g->an.f_b[s]);
  }
  {
    const Analysis& A = g->an;
    const int Bn = std::max(1, A.band_levels);
    g->stage_max_piv.assign(A.n_stages, 1);
    for (int s = 0; s < A.n_fronts; s++) { int& m = g->stage_max_piv[A.f_level[s] / Bn]; m = std::max(m, A.f_p[s]); }
    int max_piv = 0;
    for (int m : g->stage_max_piv) max_piv = std::max(max_piv, m);
    g->use_band = A.max_front <= band_front_limit() && max_piv <= 64;
    // the dense-front solve keeps a front's boundary values in LDS: 15 000 scalars is the ceiling (2-D loop-closure
    // meshes such as torus10000 reach 24 540 under this chain-based dissection and are refused, see below)
    g->use_dense = !g->use_band && max_piv <= dense_front_max_pivots() && A.max_front <= 15000;
    g->level_max_b.assign(A.n_levels, 0);
    g->max_el_per_front = 0;
    for (int s = 0; s < A.n_fronts; s++) {
      g->level_max_b[A.f_level[s]] = std::max(g->level_max_b[A.f_level[s]], A.f_b[s]);
      g->max_el_per_front = std::max(g->max_el_per_front, A.f_el_off[s + 1] - A.f_el_off[s]);
    }
    g->dw_asm.clear(); g->dw_pan.clear(); g->dw_trl.clear();
    if (g->use_dense)
      for (int l = 0; l < A.n_levels; l++) {
        int a = 0, pn = 0, t = 0;
        g->dw_asm.push_back(0); g->dw_pan.push_back(0); g->dw_trl.push_back(0);
        for (int k = A.level_off[l]; k < A.level_off[l + 1]; k++) {
          const int s = A.level_fronts[k];
          const int fa = A.f_p[s] + A.f_b[s] + 1, b1 = A.f_b[s] + 1;
          const int T32 = (fa + 31) / 32, T64 = (b1 + 63) / 64;
          a += T32 * ((A.f_p[s] + 31) / 32); pn += (fa - A.f_p[s] + 255) / 256; t += T64 * (T64 + 1) / 2;
          g->dw_asm.push_back(a); g->dw_pan.push_back(pn); g->dw_trl.push_back(t);
        }
      }
    g->stage_nw_factor.assign(A.n_stages, 1); g->stage_nw_solve.assign(A.n_stages, 1);
    g->stage_max_grp_fronts.assign(A.n_stages, 1); g->stage_max_panel.assign(A.n_stages, 1);
    for (int s = 0; s < A.n_fronts; s++) { int& m = g->stage_max_panel[A.f_level[s] / Bn]; m = std::max(m, (A.f_p[s] + A.f_b[s] + 1) * A.f_p[s]); }
    const size_t lds_budget = 150 * 1024;
    const int max_waves = 8;
    for (int st = 0; st < A.n_stages; st++) {
      int mg = 1;
      for (int gi = A.stage_grp_off[st]; gi < A.stage_grp_off[st + 1]; gi++)
        mg = std::max(mg, A.glvl_front_off[A.grp_lvl_off[gi + 1]] - A.glvl_front_off[A.grp_lvl_off[gi]]);
      g->stage_max_grp_fronts[st] = mg;
      // waves per group: one per front of its widest level -- or, when the whole group fits (4 + 2 + 1 on eight waves), one per front
      // of the group: every upper front is then assembled ahead of its level by a wave of its own (body_band_factor_pre, shape B)
      const bool reg_stage = A.stage_max_front[st] + 1 <= band_reg_rows() && g->sw.trace == 0;
      const int want = std::max(1, std::min(max_waves, reg_stage && mg <= max_waves && !g->sw.no_preassemble ? mg : A.stage_max_width[st]));
      g->stage_nw_factor[st] = (int)std::max<size_t>(1, std::min<size_t>(want, lds_budget / band_lds_bytes(A.stage_max_front[st], reg_stage)));
      // per workgroup: one local solution vector per front of a group + per wave xb and the factor panel.  A group that does
      // not fit (very wide elimination trees: hundreds of fronts in one band group) takes the graph off the band kernels.
      const size_t xbytes = (size_t)mg * band_max_rows() * sizeof(double);
      const size_t per_wave = band_solve_lds_bytes(g->stage_max_panel[st]);
      if (xbytes + per_wave > lds_budget) { g->use_band = false; g->stage_nw_solve[st] = 1; continue; }
      g->stage_nw_solve[st] = (int)std::max<size_t>(1, std::min<size_t>(want, (lds_budget - xbytes) / per_wave));
    }
    // k_band_factor_pre: every level in one round of the stage's waves, and either the whole group on waves of its own (shape B: 4 + 2 + 1
    // on eight waves) or three to four local levels with the fronts of local levels 2 and up on waves that idle from level 1 on (shape A:
    // 8 + 4 + 2 + 1 on eight waves, 4 + 2 + 1 <= 8) -- body_band_factor_pre tells the two apart the same way
    g->stage_pre.assign(A.n_stages, 0);
    if (!g->sw.no_preassemble)
      for (int st = 0; st < A.n_stages; st++) {
        bool ok = A.stage_grp_off[st + 1] > A.stage_grp_off[st] && A.stage_max_front[st] + 1 <= band_reg_rows();
        const int nw = g->stage_nw_factor[st];
        for (int gi = A.stage_grp_off[st]; ok && gi < A.stage_grp_off[st + 1]; gi++) {
          const int l0 = A.grp_lvl_off[gi], nl = A.grp_lvl_off[gi + 1] - l0;
          ok = nl >= 1 && nl <= 4;
          int upper = 0, c0 = 0;
          for (int k = 0; ok && k < nl; k++) {
            const int c = A.glvl_front_off[l0 + k + 1] - A.glvl_front_off[l0 + k];
            if (k == 0) c0 = c; else upper += c;
          }
          ok = ok && (c0 + upper <= nw || (nl >= 3 && c0 <= nw && upper <= nw));
        }
        g->stage_pre[st] = ok ? 1 : 0;
      }
    // The whole tree in one factor launch + one back-substitution launch (k_band_factor_all / k_band_solve_all, round 6): every stage takes the
    // pre-assembling walk on register-resident fronts, the data-flow back-substitution holds a group of any stage, and the groups fit the chip
    // a few times over -- a graph of thousands of groups (C3) keeps a launch per stage: its upper workgroups would sit on wave slots its leaves
    // need.  Measured over corridor graphs of 500 .. 3 500 poses (tools/size_probe.py, same box, us per LM iteration, whole tree | a launch per
    // band): 75 groups and fewer (up to 1 250 poses; C2 and every frame of C5: 73) 52.2 | 55.9, 56.5 | 57.3, 82.0 | 88.3; 127 groups and
    // more 87.7 | 77.0, 81.4 | 71.0, 85.5 | 74.9, 140.8 | 110.2 (every group's top front releases at agent scope, every waiting group acquires:
    // hundreds of L2 write-backs and invalidations inside one launch instead of one per band): at most 100 groups (it was 512).
    g->k3_all = false;
    if (g->use_band && A.n_stages >= 2 && g->sw.trace == 0 && !g->sw.no_preassemble && !g->sw.no_solve_flow && !g->sw.no_root_fuse && A.n_groups <= 100) {
      bool ok = true;
      int nwf = 1, mf = 0, mp = 1, mg = 1;
      for (int st = 0; st < A.n_stages; st++) {
        ok = ok && g->stage_pre[st] != 0 && A.stage_max_front[st] + 1 <= band_reg_rows();
        nwf = std::max(nwf, g->stage_nw_factor[st]); mf = std::max(mf, A.stage_max_front[st]);
        mp = std::max(mp, g->stage_max_panel[st]); mg = std::max(mg, g->stage_max_grp_fronts[st]);
      }
      const int nws = ok ? band_all_solve_waves(mp, mg) : 0;
      ok = ok && mg > 1 && nws >= std::min(8, mg) && band_lds_bytes(mf, true) * (size_t)nwf <= lds_budget;
      if (ok) { g->k3_all = true; g->k3_nw_factor = nwf; g->k3_nw_solve = nws; g->k3_max_front = mf; g->k3_max_panel = mp; g->k3_max_grp = mg; }
    }
    // (such a graph then runs on the level-per-launch kernels: its fronts are <= 127 rows by the use_band test above)
  }
  if (!g->use_band && !g->use_dense && g->an.max_front > 4096)
    return fail(g, PPS_ENOMEM, "fronts too wide for this ordering (max front " + std::to_string(g->an.max_front) +
                               " scalars): the pose chain is not a good dissection backbone for this graph");
  if (g->sw.analysis_timing) fprintf(stderr, "[analysis] %-22s %8.3f ms\n", "total incl. api", 1e3 * (now_s() - t0));
  g->analyzed = true; g->analysis_stale = false;
  g->n_analyses++; g->grown_only = true;
  g->an_base_seq = g->an_seq; g->an_seq++;                // (an.kept is relative to the analysis this one replaced)
  g->stats.n_fronts = g->an.n_fronts; g->stats.n_levels = g->an.n_levels; g->stats.max_front = g->an.max_front;
  g->stats.nnz_L = g->an.L_size;
  g->stats.t_analysis = now_s() - t0;
  return PPS_OK;
}

// pinned staging for the estimate (pose rows, then plane rows): copies to and from pageable memory are staged by the runtime
// and cost a synchronisation each
int state_pin_reserve(pps_graph* g, size_t doubles) {
  if (doubles <= g->state_pin_cap) return PPS_OK;
  if (g->up_inflight) { HIP_TRY(g, hipStreamSynchronize(g->stream)); g->up_inflight = false; }
  if (g->state_pin) (void)hipHostFree(g->state_pin);
  g->state_pin = nullptr; g->state_pin_cap = 0; g->pin_holds_est = false;
  const size_t cap = std::max<size_t>(4096, 2 * doubles);
  HIP_TRY(g, hipHostMalloc(reinterpret_cast<void**>(&g->state_pin), cap * sizeof(double), hipHostMallocDefault));
  g->state_pin_cap = cap;
  return PPS_OK;
}

// pull the device estimate back into the host node table
// The solves end with this: the estimate travels to the pinned buffer behind their last kernels and arrives with the
// synchronisation they end with anyway -- whoever reads a value next (a frame loop does, every frame) finds it on the host
// instead of paying a copy and a synchronisation of its own.  Call before that synchronisation, state_download_arrived() after it.
int enqueue_state_download(pps_graph* g) {
  const DevGraph& d = g->dev;
  const size_t np = (size_t)7 * d.pose_ld, nl = (size_t)4 * d.plane_ld;
  g->pin_holds_est = false;
  int rc = state_pin_reserve(g, np + nl);
  if (rc != PPS_OK) return rc;
  HIP_TRY(g, hipMemcpyAsync(g->state_pin, d.pose_est, (np + nl) * 8, hipMemcpyDeviceToHost, g->stream));   // [poses | planes], one block
  return PPS_OK;
}
void state_download_arrived(pps_graph* g) { g->pin_holds_est = true; }

int download_state(pps_graph* g) {
  if (!g->dev_values_newer) return PPS_OK;
  HIP_TRY(g, hipSetDevice(g->props.device));
  const DevGraph& d = g->dev;
  const size_t np = (size_t)7 * d.pose_ld, nl = (size_t)4 * d.plane_ld;
  if (!g->pin_holds_est) {
    int rc = state_pin_reserve(g, np + nl);
    if (rc != PPS_OK) return rc;
    HIP_TRY(g, hipMemcpyAsync(g->state_pin, d.pose_est, (np + nl) * 8, hipMemcpyDeviceToHost, g->stream));   // [poses | planes], one block
    HIP_TRY(g, hipStreamSynchronize(g->stream));
  }
  g->pin_holds_est = false;
  double* bp = g->state_pin; double* bl = g->state_pin + np;
  for (int s = 0; s < d.n_pose; s++) for (int k = 0; k < 7; k++) g->nodes[g->pose_ids[s]].v[k] = bp[(size_t)k * d.pose_ld + s];
  for (int s = 0; s < d.n_plane; s++) for (int k = 0; k < 4; k++) g->nodes[g->plane_ids[s]].v[k] = bl[(size_t)k * d.plane_ld + s];
  g->dev_values_newer = false;
  return PPS_OK;
}

int upload_state(pps_graph* g, bool sync) {
  DevGraph& d = g->dev;
  if (g->up_inflight) { HIP_TRY(g, hipStreamSynchronize(g->stream)); g->up_inflight = false; }   // the staging buffer is still being read
  const size_t np = (size_t)7 * d.pose_ld, nl = (size_t)4 * d.plane_ld;
  int rc = state_pin_reserve(g, np + nl);
  if (rc != PPS_OK) return rc;
  double* bp = g->state_pin; double* bl = g->state_pin + np;
  g->pin_holds_est = false;
  for (int k = 0; k < 7; k++) for (int s = d.n_pose; s < d.pose_ld; s++) bp[(size_t)k * d.pose_ld + s] = 0.0;
  for (int k = 0; k < 4; k++) for (int s = d.n_plane; s < d.plane_ld; s++) bl[(size_t)k * d.plane_ld + s] = 0.0;
  for (int s = 0; s < d.n_pose; s++) for (int k = 0; k < 7; k++) bp[(size_t)k * d.pose_ld + s] = g->nodes[g->pose_ids[s]].v[k];
  for (int s = 0; s < d.n_plane; s++) for (int k = 0; k < 4; k++) bl[(size_t)k * d.plane_ld + s] = g->nodes[g->plane_ids[s]].v[k];
  HIP_TRY(g, hipMemcpyAsync(d.pose_est, bp, (np + nl) * 8, hipMemcpyHostToDevice, g->stream));
  // (the linearisation point is set by whoever linearises next: copy_state / linpoint_from_estimate)
  if (sync) HIP_TRY(g, hipStreamSynchronize(g->stream));
  else g->up_inflight = true;
  g->host_values_newer = false;
  return PPS_OK;
}

// SoA with leading dimension ld (>= count): value k of slot s at [k * ld + s]
template <int K>
void pack_soa(const pps_graph* g, int type, const double HostFactor::*dummy, bool weights, std::vector<double>& out, size_t ld) {
  (void)dummy;
  const std::vector<int>& ids = g->fslot_ids[type];
  const size_t n = ids.size();
  out.assign((size_t)K * ld, 0.0);
  for (size_t s = 0; s < n; s++) {
    const HostFactor& f = g->factors[ids[s]];
    const double* src = weights ? f.w : f.meas;
    for (int k = 0; k < K; k++) out[(size_t)k * ld + s] = src[k];
  }
}

// pull device-refreshed plane-observation measurements back into the host factor table
int download_measurements(pps_graph* g) {
  if (!g->dev_meas_newer) return PPS_OK;
  HIP_TRY(g, hipSetDevice(g->props.device));
  const DevGraph& d = g->dev;
  const size_t n = g->fslot_ids[F_PLANE_OBS].size(), ld = (size_t)d.obs_ld;
  std::vector<double> m((size_t)4 * ld);
  if (n) HIP_TRY(g, hipMemcpyAsync(m.data(), d.obs_meas, m.size() * 8, hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  for (size_t s2 = 0; s2 < n && s2 < (size_t)d.n_obs; s2++) {
    HostFactor& f = g->factors[g->fslot_ids[F_PLANE_OBS][s2]];
    for (int k = 0; k < 4; k++) f.meas[k] = m[(size_t)k * ld + s2];
  }
  g->dev_meas_newer = false;
  g->pk_meas_ok = false;
  return PPS_OK;
}

int upload_measurements(pps_graph* g) {
  DevGraph& d = g->dev;
  std::vector<double> m;
  pack_soa<4>(g, F_PLANE_OBS, nullptr, false, m, (size_t)d.obs_ld);
  if (d.n_obs) HIP_TRY(g, hipMemcpyAsync(d.obs_meas, m.data(), m.size() * 8, hipMemcpyHostToDevice, g->stream));
  std::vector<double> m2;
  pack_soa<4>(g, F_PLANE_PRIOR, nullptr, false, m2, (size_t)d.lp_ld);
  if (d.n_lp) HIP_TRY(g, hipMemcpyAsync(d.lp_meas, m2.data(), m2.size() * 8, hipMemcpyHostToDevice, g->stream));
  // (the upload mirror no longer describes these arrays: the next topology upload sends them whole)
  g->up_unknown_meas = true;
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  g->meas_dirty = false;
  return PPS_OK;
}

int upload_all(pps_graph* g) {
  const double t0 = now_s();
  const bool was_grown_only = g->grown_only_upload;
  const bool tm = g->sw.upload_timing;
  g->verify_hints = g->sw.verify_upload;
  double tl = t0;
  auto lap = [&](const char* what) { if (!tm) return; const double t = now_s(); g->up_laps[what] += t - tl; tl = t; };
  int rc = ensure_device(g);
  if (rc != PPS_OK) return rc;
  lap("0 ensure_device");
  if (g->dev_values_newer) { rc = download_state(g); if (rc != PPS_OK) return rc; }
  lap("1a state download");
  // Refreshed measurements may stay on the device across an upload that only appends (see the obs_meas upload below): same
  // arena, same slot with room for the new rows, same leading dimension, no re-popping edges (their slots sit behind the
  // fixed ones and would move).
  bool keep_meas = false;
  const size_t n_obs_on_device = (size_t)g->dev.n_obs;         // (free_device below resets g->dev)
  if (g->dev_meas_newer) {
    // (live counts kept by the add / remove calls: a walk over the 300-byte factor records here was 30 us per frame at 7 600 factors)
    const size_t n_obs_new = (size_t)g->n_live_type[F_PLANE_OBS], n_lp_new = (size_t)g->n_live_type[F_PLANE_PRIOR];
    const bool any_repop = g->n_live_repop > 0;
    keep_meas = g->grown_only_upload && !g->up_unknown && !g->up_unknown_meas && g->up.spill == 0 && !any_repop && g->dev.n_obs == g->dev.n_obs_fixed &&
                j_capacity((int64_t)n_obs_new) == g->dev.obs_ld && j_capacity((int64_t)n_lp_new) == g->dev.lp_ld && g->slot_obs_meas < g->up_slots.size() &&
                true;
    if (!keep_meas) {
      if (tm) g->up_laps["(measurement downloads, count)"] += 1e-3;      // (printed as ms: the count)
      rc = download_measurements(g); if (rc != PPS_OK) return rc;
    }
  }
  lap("1b measurements");
  // the analysis is host work on host tables: it runs while the device finishes what the last frame left on the stream (the measurement
  // refresh of a frame loop: the opening sync below waited 10 us per frame for it)
  if (!g->analyzed || g->analysis_stale) { rc = run_analysis(g); if (rc != PPS_OK) return rc; }
  lap("3 analysis");
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  g->up_inflight = false;
  lap("1c opening stream sync");
  free_device(g);
  g->spec_L = g->spec_U = g->spec_delta = nullptr; g->spec_result = nullptr;
  g->spec_pose = g->spec_plane = g->spec_chi2_partials = g->spec_dn_partials = nullptr; g->spec_ticket = nullptr;
  g->spec_J = g->spec_P = g->spec_H = g->spec_Hf = nullptr;
  g->d_item_frame = g->d_item_plane = g->d_item_slot = g->d_frame_pose_slot = g->d_frame_seg_off = nullptr; g->d_fr_seg = nullptr;
  g->frames_dirty = true;
  g->snap_pose = g->snap_plane = nullptr; g->upload_version++;
  lap("2 free_device");
  const Analysis& A = g->an;
  DevGraph& d = g->dev;
  // the mirror holds the arrays of the analysis this one was built upon: its kept parts are not compared again
  const bool hints = was_grown_only && !g->up_unknown && g->up_an_seq >= 0 && g->an_base_seq == g->up_an_seq;
  const Analysis::Kept K = hints ? A.kept : Analysis::Kept();
  const size_t kF = (size_t)K.fronts, kFl = (size_t)K.fronts_lists, kB = (size_t)K.blocks, kS = (size_t)K.segs;
  auto at = [](const auto& v, size_t i) -> size_t { return i < v.size() ? (size_t)v[i] : 0; };     // (0 = no claim)
  double* zero_block = nullptr; size_t zero_doubles = 0;
  d.n_pose = (int)g->pose_ids.size(); d.n_plane = (int)g->plane_ids.size();
  d.no_strip = g->sw.no_strip ? 1 : 0;
  d.sw = g->sw.dev_bits();
  HIP_TRY(g, step_constants(d.step_ac));
  d.pose_ld = std::max(1, (d.n_pose + 63) / 64 * 64); d.plane_ld = std::max(1, (d.n_plane + 63) / 64 * 64);
#define TRY(x) do { rc = (x); if (rc != PPS_OK) return rc; } while (0)
  // every copy of the state is one block [poses | planes]: one transfer moves it (copies rotate by pointer pairs, so a
  // plane array always sits behind its pose array)
  const size_t state_doubles = (size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld;
  TRY(dev_alloc(g, &d.pose_est, state_doubles)); d.plane_est = d.pose_est + (size_t)7 * d.pose_ld;
  TRY(dev_alloc(g, &d.pose_lin, state_doubles)); d.plane_lin = d.pose_lin + (size_t)7 * d.pose_ld;
  std::vector<int> pv(d.n_pose), lv(d.n_plane);
  for (int s = 0; s < d.n_pose; s++) pv[s] = A.node_voff[g->nodes[g->pose_ids[s]].compact];
  for (int s = 0; s < d.n_plane; s++) lv[s] = A.node_voff[g->nodes[g->plane_ids[s]].compact];
  TRY(dev_upload(g, &d.pose_voff, pv)); TRY(dev_upload(g, &d.plane_voff, lv));
  // factors
  d.n_obs = (int)g->fslot_ids[F_PLANE_OBS].size(); d.n_odo = (int)g->fslot_ids[F_ODOMETRY].size();
  d.n_pp = (int)g->fslot_ids[F_POSE_PRIOR].size(); d.n_lp = (int)g->fslot_ids[F_PLANE_PRIOR].size();
  { int64_t base[4]; j_bases(g, base, nullptr); d.joff_obs = base[F_PLANE_OBS]; d.joff_odo = base[F_ODOMETRY]; d.joff_pp = base[F_POSE_PRIOR]; d.joff_lp = base[F_PLANE_PRIOR]; }
  auto idx_of = [&](int type, bool second) {
    std::vector<int> v(g->fslot_ids[type].size());
    for (size_t s = 0; s < v.size(); s++) {
      const HostFactor& f = g->factors[g->fslot_ids[type][s]];
      v[s] = g->nodes[second ? f.b : f.a].slot;
    }
    return v;
  };
  std::vector<double> tmp;
  // the previous upload's packed arrays are still right for the old factors when nothing was removed since (slots only append)
  const bool incr_pack = was_grown_only;
  d.obs_ld = (int)j_capacity(d.n_obs); d.odo_ld = (int)j_capacity(d.n_odo); d.pp_ld = (int)j_capacity(d.n_pp); d.lp_ld = (int)j_capacity(d.n_lp);
  {
    // one pass over the plane observations (a HostFactor is 300 bytes: four passes were four times the memory traffic), and
    // only over the new ones when the graph has just grown
    const std::vector<int>& ids = g->fslot_ids[F_PLANE_OBS];
    const size_t n = ids.size(), ld = (size_t)d.obs_ld;
    std::vector<int>& ia = g->pk_obs_a; std::vector<int>& ib = g->pk_obs_b;
    std::vector<double>& pm = g->pk_obs_m; std::vector<double>& pw = g->pk_obs_w;
    size_t s_begin = 0;
    if (incr_pack && g->pk_ld_obs == ld && g->pk_n_obs <= n && pm.size() == 4 * ld && g->pk_obs_ids.size() == g->pk_n_obs &&
        std::equal(g->pk_obs_ids.begin(), g->pk_obs_ids.end(), ids.begin())) s_begin = g->pk_n_obs;   // (re-popping edges sit behind the fixed ones: their slots move)
    ia.resize(n); ib.resize(n); pm.resize(4 * ld); pw.resize(6 * ld);
    for (size_t s2 = s_begin; s2 < n; s2++) {
      const HostFactor& f = g->factors[ids[s2]];
      ia[s2] = g->nodes[f.a].slot; ib[s2] = g->nodes[f.b].slot;
      for (int k = 0; k < 4; k++) pm[(size_t)k * ld + s2] = f.meas[k];
      for (int k = 0; k < 6; k++) pw[(size_t)k * ld + s2] = f.w[k];
    }
    if (s_begin > 0 && !g->pk_meas_ok)                                 // the host's measurements changed: those rows again
      for (size_t s2 = 0; s2 < s_begin; s2++) { const HostFactor& f = g->factors[ids[s2]]; for (int k = 0; k < 4; k++) pm[(size_t)k * ld + s2] = f.meas[k]; }
    g->pk_n_obs = n; g->pk_ld_obs = ld; g->pk_obs_ids.resize(s_begin); g->pk_obs_ids.insert(g->pk_obs_ids.end(), ids.begin() + (std::ptrdiff_t)s_begin, ids.end());
    const size_t same_idx = hints ? s_begin : 0, same_meas = (hints && g->pk_meas_ok) ? s_begin : 0;       // (rows packed and sent by the previous upload)
    TRY(dev_upload(g, &d.obs_pose, ia, same_idx)); TRY(dev_upload(g, &d.obs_plane, ib, same_idx));
    // Measurements that a device-side refresh has rewritten (pps_refresh_measurements) stay where they are when this upload
    // only appends: the host packs its (older) copies, the mirror holds the same bytes, so nothing is sent for them and the
    // device keeps the refreshed values; only the new observations travel.  dev_meas_newer stays set.
    g->slot_obs_meas = g->up_cursor;
    TRY(dev_upload_rows(g, &d.obs_meas, pm, 4, ld, n, g->up_unknown_meas, keep_meas ? std::min(n, n_obs_on_device) : kNoExact, same_meas));
    TRY(dev_upload_rows(g, &d.obs_w, pw, 6, ld, n, false, kNoExact, same_idx));
  }
  d.n_obs_fixed = g->n_obs_fixed;
  {
    const std::vector<int>& ids = g->fslot_ids[F_PLANE_OBS];
    const size_t n2 = ids.size() - (size_t)g->n_obs_fixed;
    if (n2 > 0) {
      tmp.assign(6 * n2, 0.0);
      for (size_t k = 0; k < n2; k++)
        for (int c = 0; c < 6; c++) tmp[(size_t)c * n2 + k] = g->factors[ids[g->n_obs_fixed + k]].ray[c];
      TRY(dev_upload(g, &d.obs_ray, tmp));
    }
  }
  {
    const std::vector<int>& ids = g->fslot_ids[F_ODOMETRY];
    const size_t n = ids.size(), ld = (size_t)d.odo_ld;
    std::vector<int>& ia = g->pk_odo_a; std::vector<int>& ib = g->pk_odo_b;
    std::vector<double>& pm = g->pk_odo_m; std::vector<double>& pw = g->pk_odo_w;
    size_t s_begin = 0;
    if (incr_pack && g->pk_ld_odo == ld && g->pk_n_odo <= n && pm.size() == 6 * ld && g->pk_odo_ids.size() == g->pk_n_odo &&
        std::equal(g->pk_odo_ids.begin(), g->pk_odo_ids.end(), ids.begin())) s_begin = g->pk_n_odo;
    ia.resize(n); ib.resize(n); pm.resize(6 * ld); pw.resize(21 * ld);
    for (size_t s2 = s_begin; s2 < n; s2++) {
      const HostFactor& f = g->factors[ids[s2]];
      ia[s2] = g->nodes[f.a].slot; ib[s2] = g->nodes[f.b].slot;
      for (int k = 0; k < 6; k++) pm[(size_t)k * ld + s2] = f.meas[k];
      for (int k = 0; k < 21; k++) pw[(size_t)k * ld + s2] = f.w[k];
    }
    g->pk_n_odo = n; g->pk_ld_odo = ld; g->pk_odo_ids.resize(s_begin); g->pk_odo_ids.insert(g->pk_odo_ids.end(), ids.begin() + (std::ptrdiff_t)s_begin, ids.end());
    const size_t same_idx = hints ? s_begin : 0;
    TRY(dev_upload(g, &d.odo_a, ia, same_idx)); TRY(dev_upload(g, &d.odo_b, ib, same_idx));
    TRY(dev_upload_rows(g, &d.odo_meas, pm, 6, ld, n, false, kNoExact, (hints && g->pk_meas_ok) ? s_begin : 0));
    TRY(dev_upload_rows(g, &d.odo_w, pw, 21, ld, n, false, kNoExact, same_idx));
  }
  TRY(dev_upload(g, &d.pp_pose, idx_of(F_POSE_PRIOR, false)));
  pack_soa<6>(g, F_POSE_PRIOR, nullptr, false, tmp, (size_t)d.pp_ld); TRY(dev_upload_rows(g, &d.pp_meas, tmp, 6, (size_t)d.pp_ld, (size_t)d.n_pp, false));
  pack_soa<21>(g, F_POSE_PRIOR, nullptr, true, tmp, (size_t)d.pp_ld); TRY(dev_upload_rows(g, &d.pp_w, tmp, 21, (size_t)d.pp_ld, (size_t)d.n_pp, false));
  TRY(dev_upload(g, &d.lp_plane, idx_of(F_PLANE_PRIOR, false)));
  pack_soa<4>(g, F_PLANE_PRIOR, nullptr, false, tmp, (size_t)d.lp_ld); g->slot_lp_meas = g->up_cursor; TRY(dev_upload_rows(g, &d.lp_meas, tmp, 4, (size_t)d.lp_ld, (size_t)d.n_lp, g->up_unknown_meas));
  pack_soa<6>(g, F_PLANE_PRIOR, nullptr, true, tmp, (size_t)d.lp_ld); TRY(dev_upload_rows(g, &d.lp_w, tmp, 6, (size_t)d.lp_ld, (size_t)d.n_lp, false));
  g->up_unknown_meas = false;
  g->pk_meas_ok = true;
  lap("4 factor packing + diff");
  // linear system storage
  TRY(dev_alloc(g, &d.J, (size_t)A.J_size)); TRY(dev_alloc(g, &d.H, (size_t)A.H_size));
  TRY(dev_alloc(g, &d.P, (size_t)std::max<int64_t>(1, A.P_size)));
  { int64_t pb[4]; p_bases(g, pb, nullptr); d.poff_obs = pb[F_PLANE_OBS]; }
  TRY(dev_alloc(g, &d.L, (size_t)A.L_size)); TRY(dev_alloc(g, &d.U, (size_t)A.U_size));
  TRY(dev_alloc(g, &g->spec_L, (size_t)A.L_size)); TRY(dev_alloc(g, &g->spec_U, (size_t)A.U_size));
  const size_t delta_doubles = (size_t)std::max(1, A.n_scalars);   // (delta and the second delta sit in the zeroed block below)
  d.n_scalars = A.n_scalars;
  d.n_fronts = A.n_fronts; d.n_levels = A.n_levels; d.max_front = A.max_front; d.n_segs = A.n_segs; d.n_blocks = A.n_blocks;
  TRY(dev_upload(g, &d.f_p, A.f_p, kF)); TRY(dev_upload(g, &d.f_b, A.f_b, kF)); TRY(dev_upload(g, &d.f_poff, A.f_poff, kF)); TRY(dev_upload(g, &d.pidx, A.pidx, kF ? at(A.f_poff, kF) : 0));
  TRY(dev_upload(g, &d.f_Loff, A.f_Loff, kF)); TRY(dev_upload(g, &d.f_Uoff, A.f_Uoff, kF));
  TRY(dev_upload(g, &d.f_bidx_off, A.f_bidx_off, kF ? kF + 1 : 0)); TRY(dev_upload(g, &d.bidx, A.bidx, kF ? at(A.f_bidx_off, kF) : 0));
  TRY(dev_upload(g, &d.f_child_off, A.f_child_off, kF ? kF + 1 : 0)); TRY(dev_upload(g, &d.child, A.child, kF ? at(A.f_child_off, kF) : 0));
  TRY(dev_upload(g, &d.f_cmap_off, A.f_cmap_off, kF ? kF + 1 : 0)); TRY(dev_upload(g, &d.cmap, A.cmap));      // (cmap: kept fronts below a redone parent are rewritten)
  TRY(dev_upload(g, &d.level_fronts, A.level_fronts));
  const size_t kA = kFl ? at(A.f_asm_off, kFl) : 0;
  TRY(dev_upload(g, &d.f_asm_off, A.f_asm_off, kFl ? kFl + 1 : 0)); TRY(dev_upload(g, &d.asm_blk, A.asm_blk, kA));
  TRY(dev_upload(g, &d.asm_lrow, A.asm_lrow, kA)); TRY(dev_upload(g, &d.asm_lcol, A.asm_lcol, kA));
  TRY(dev_upload(g, &d.blk_rows, A.blk_rows, kB)); TRY(dev_upload(g, &d.blk_cols, A.blk_cols, kB));
  TRY(dev_upload(g, &d.blk_size, A.blk_size, kB)); TRY(dev_upload(g, &d.blk_nseg, A.blk_nseg, kB));
  TRY(dev_upload(g, &d.blk_hoff, A.blk_hoff, kB));
  TRY(dev_upload(g, &d.seg_blk, A.seg_blk, kS)); TRY(dev_upload(g, &d.seg_c0, A.seg_c0, kS)); TRY(dev_upload(g, &d.seg_cnt, A.seg_cnt, kS));
  TRY(dev_upload(g, &d.seg_hoff, A.seg_hoff, kS));
  TRY(dev_upload(g, &d.contrib, A.contrib, 4 * (size_t)K.contribs));
  {
    std::vector<int> mseg;
    for (int bk = 0; bk < A.n_blocks; bk++) if (A.blk_nseg[bk] > 1) mseg.push_back(bk);
    d.n_mseg = (int)mseg.size();
    TRY(dev_upload(g, &d.mseg_blk, mseg));
    // K2's two work lists: segments of single-segment blocks that K1 does not write itself; first segment of every other block
    std::vector<int> k2s, k2m, k2f;
    for (int sg : A.nd_segs) if (A.blk_nseg[A.seg_blk[sg]] == 1) k2s.push_back(sg);
    for (int sg = 0; sg < A.n_segs; sg++) {
      const int ns = A.blk_nseg[A.seg_blk[sg]];
      if (ns <= 1 || (sg > 0 && A.seg_blk[sg - 1] == A.seg_blk[sg])) continue;      // (first segment of a block of several)
      for (int c0 = 0; c0 < ns; c0 += 16) { k2m.push_back(sg + c0); k2m.push_back(std::min(16, ns - c0) | (ns <= 16 ? 1 << 16 : 0)); }
      if (ns > 16) k2f.push_back(sg);
    }
    d.n_k2_single = (int)k2s.size(); d.n_k2_multi = (int)k2m.size() / 2; d.n_k2_finish = (int)k2f.size();
    TRY(dev_upload(g, &d.k2_single, k2s)); TRY(dev_upload(g, &d.k2_multi, k2m)); TRY(dev_upload(g, &d.k2_finish, k2f));
  }
  TRY(dev_upload(g, &d.f_el_off, A.f_el_off, kFl ? kFl + 1 : 0)); TRY(dev_alloc(g, &d.el_tgt, (size_t)std::max<int64_t>(1, A.el_total)));   // filled by k_expand_el below
  TRY(dev_upload(g, &d.asm_el0, A.asm_el0, kA)); TRY(dev_upload(g, &d.asm_fsz, A.asm_fsz, kA));
  TRY(dev_upload(g, &d.f_ea_off, A.f_ea_off, kF ? kF + 1 : 0)); TRY(dev_alloc(g, &d.ea_tgt, (size_t)std::max<int64_t>(1, A.ea_total)));   // filled by k_expand_ea below
  TRY(dev_upload(g, &d.blk_doff, A.blk_doff, kB ? kB + 1 : 0)); TRY(dev_alloc(g, &d.blk_dst, (size_t)std::max(1, A.blk_doff[A.n_blocks])));
  TRY(dev_alloc(g, &d.Hf, (size_t)A.el_total));
  if (g->use_band && !g->sw.no_spec_lin) {                        // the spare set of the fused trial + linearisation launch (lm_solve_dual)
    TRY(dev_alloc(g, &g->spec_J, (size_t)A.J_size)); TRY(dev_alloc(g, &g->spec_P, (size_t)std::max<int64_t>(1, A.P_size)));
    TRY(dev_alloc(g, &g->spec_H, (size_t)A.H_size)); TRY(dev_alloc(g, &g->spec_Hf, (size_t)A.el_total));
  }
  TRY(dev_upload(g, &d.grp_lvl_off, A.grp_lvl_off)); TRY(dev_upload(g, &d.glvl_front_off, A.glvl_front_off));
  {
    // per group, one 32-byte record: first position, fronts, local levels, first position of local levels 1 .. 4 (the last one
    // repeated), 0 -- what the band kernels would otherwise fetch with two dependent look-ups (levels of the group, their offsets)
    std::vector<int> span(8 * (size_t)std::max(1, A.n_groups), 0);
    for (int gi = 0; gi < A.n_groups; gi++) {
      const int l0 = A.grp_lvl_off[gi], nl = A.grp_lvl_off[gi + 1] - l0;
      int* r = &span[8 * (size_t)gi];
      r[0] = A.glvl_front_off[l0]; r[1] = A.glvl_front_off[l0 + nl] - r[0]; r[2] = nl;
      for (int k = 1; k <= 4; k++) r[2 + k] = A.glvl_front_off[l0 + std::min(k, nl)];
    }
    TRY(dev_upload(g, &d.grp_span, span));
  }
  TRY(dev_upload(g, &d.glvl_fronts, A.glvl_fronts));
  TRY(dev_upload(g, &d.frec, A.frec)); TRY(dev_upload(g, &d.crec, A.crec)); TRY(dev_upload(g, &d.srec, A.srec, 8 * kS));
  TRY(dev_upload(g, &d.obs_dir, A.obs_dir)); TRY(dev_upload(g, &d.nd_segs, A.nd_segs, (size_t)K.nd_segs)); d.n_nd_segs = (int)A.nd_segs.size();
  TRY(dev_upload(g, &d.cls_off, A.cls_off)); TRY(dev_upload(g, &d.cls_fronts, A.cls_fronts));
  if (g->use_dense) { TRY(dev_upload(g, &g->d_dw_asm, g->dw_asm)); TRY(dev_upload(g, &g->d_dw_pan, g->dw_pan)); TRY(dev_upload(g, &g->d_dw_trl, g->dw_trl)); }
  d.chi2_blocks = (d.n_obs + 255) / 256 + (d.n_odo + 255) / 256 + (d.n_pp + 255) / 256 + (d.n_lp + 255) / 256;
  TRY(dev_alloc(g, &d.chi2_partials, (size_t)std::max(1, d.chi2_blocks)));

  {
    // one zeroed block: [dn_partials | ticket | spec ticket | result record | the second factorisation's result record |
    // delta | the second delta | the hand-over flags of the whole-tree launches: 4 ints per front]
    const size_t n_dn = (size_t)(d.n_pose + d.n_plane + 255) / 256 + 1;
    const size_t flag_doubles = 2 * (size_t)std::max(1, A.n_fronts) + 1;
    double* zb = nullptr;
    TRY(dev_alloc(g, &zb, n_dn + 2 + 8 + 2 * delta_doubles + flag_doubles));
    zero_block = zb; zero_doubles = n_dn + 2 + 8 + 2 * delta_doubles + flag_doubles;     // (cleared by k_expand_ea below, or by a memset when that launch has nothing to expand)
    d.k3_flag = reinterpret_cast<int*>(zb + n_dn + 2 + 8 + 2 * delta_doubles);
    g->k3_epoch = 0;
    d.delta = zb + n_dn + 10; g->spec_delta = d.delta + delta_doubles;
    d.dn_partials = zb;
    d.ticket = reinterpret_cast<unsigned int*>(zb + n_dn);
    g->spec_ticket = reinterpret_cast<unsigned int*>(zb + n_dn + 1);
    d.result_dev = zb + n_dn + 2; g->spec_result = zb + n_dn + 6;
    g->status_clean = true;
  }
  TRY(dev_alloc(g, &g->spec_pose, state_doubles + 1)); g->spec_plane = g->spec_pose + (size_t)7 * d.pose_ld;
  TRY(dev_alloc(g, &g->spec_chi2_partials, (size_t)std::max(1, d.chi2_blocks)));
  TRY(dev_alloc(g, &g->spec_dn_partials, (size_t)(d.n_pose + d.n_plane + 255) / 256 + 1));
  d.trace_solve = g->sw.trace >= 2 ? 1 : 0;
  if (g->sw.trace) { TRY(dev_alloc(g, &d.trace, (size_t)A.n_fronts * 8)); HIP_TRY(g, hipMemset(d.trace, 0, (size_t)A.n_fronts * 64)); }
  // fronts that exceed the LDS limit run from a global workspace (one slab per front of the widest level)
  if (!g->use_band && !g->use_dense && A.max_front > lds_front_limit()) {
    const int fa = A.max_front + 1;
    d.gwork_stride = (int64_t)fa * (fa | 1);
    int widest = 0;
    for (int l = 0; l < A.n_levels; l++)
      if (g->level_max_front[l] > lds_front_limit()) widest = std::max(widest, A.level_off[l + 1] - A.level_off[l]);
    TRY(dev_alloc(g, &d.gwork, (size_t)d.gwork_stride * std::max(1, widest)));
  }
#undef TRY
  lap("5 index arrays + diff");
  rc = flush_uploads(g); if (rc != PPS_OK) return rc;
  rc = verify_uploads(g, "upload_all"); if (rc != PPS_OK) return rc;
  lap("6 flush");
  // one launch for both expansions when every H block is assembled by exactly one front (always, unless an analysis ever lists a block
  // twice or not at all: then the fill of blk_dst has to run first, in a launch of its own)
  const bool one_launch = A.ea_total > 0 && zero_doubles < (size_t)1 << 30 && (int)A.asm_blk.size() == A.n_blocks && A.n_fronts > 0 &&
                          !g->sw.split_expand;
  if (one_launch) HIP_TRY(g, launch_expand_lists(d, A.n_fronts, (int)A.asm_blk.size(), zero_block, zero_doubles, g->stream));
  else {
    const size_t n_dst = (size_t)std::max(1, A.blk_doff[A.n_blocks]);
    if (A.ea_total > 0 && zero_doubles < (size_t)1 << 30 && n_dst < (size_t)1 << 30)
      HIP_TRY(g, launch_expand_ea(d, A.n_fronts, zero_block, zero_doubles, d.blk_dst, n_dst, g->stream));
    else {
      if (A.ea_total > 0) HIP_TRY(g, launch_expand_ea(d, A.n_fronts, nullptr, 0, nullptr, 0, g->stream));
      HIP_TRY(g, hipMemsetAsync(zero_block, 0, zero_doubles * 8, g->stream));
      HIP_TRY(g, hipMemsetAsync(d.blk_dst, 0xff, sizeof(int) * n_dst, g->stream));
    }
    HIP_TRY(g, launch_expand_el(d, (int)A.asm_blk.size(), g->stream));
  }
  g->topo_dirty = false;
  g->meas_dirty = false;
  g->grown_only_upload = true;
  g->up_an_seq = g->an_seq;
  lap("7 expand kernels + sync");
  // no sync: the solve that follows runs on the same stream (and ends with one); whatever writes the pinned buffers
  // again checks up_inflight or follows upload_all's opening sync
  rc = upload_state(g, false);
  lap("8 upload_state");
  g->stats.t_upload = now_s() - t0 - g->stats.t_analysis;
  return rc;
}

int prepare_solve(pps_graph* g) {
  int rc;
  if (g->n_live_nodes == 0) return fail(g, PPS_ESTATE, "empty graph");
  if (g->dev_ready) HIP_TRY(g, hipSetDevice(g->props.device));   // handles may be driven from any host thread
  if (g->topo_dirty || !g->dev_ready || g->dev.n_scalars == 0) { rc = upload_all(g); if (rc != PPS_OK) return rc; }
  if (g->host_values_newer) { rc = upload_state(g); if (rc != PPS_OK) return rc; }
  if (g->meas_dirty) { rc = upload_measurements(g); if (rc != PPS_OK) return rc; }
  return PPS_OK;
}

}  // namespace pps_impl
