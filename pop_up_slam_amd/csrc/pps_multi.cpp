// pps_multi.cpp
// =========================================================================================
// pps_multi: G independent graphs solved side by side.  One C2-size solve is a dependency chain that keeps a few dozen
// of the 256 CUs busy; here every kernel of an LM trial is launched ONCE for all graphs (blockIdx.y = graph) and the
// graphs advance in lockstep rounds -- a round = [re-linearise the graphs whose last trial was accepted] + factor +
// solve + trial step + chi2 for every graph still iterating.  Per-graph lambda / accept / reject run on the host from
// one 32-byte record per graph and round, with exactly the control flow (and the arithmetic) of pps_batch_optimize.
// =========================================================================================
#include "pps_graph.h"

using namespace pps;
using namespace pps_impl;

extern "C" {

struct pps_multi {
  std::vector<pps_graph*> gs;
  Switches sw;                     // the PPS_* environment switches as they were at pps_multi_create
  int n_chunks_last = 0, forms_last = 0;   // of the last solve: chunks the batch was cut into; bit 0 thread-per-factor K1 + class-body K2, bit 1 level-per-launch K3 (any chunk)
  int device = 0;
  std::string err;
  hipStream_t stream = nullptr;
  hipStream_t stream3 = nullptr;
  hipStream_t stream2 = nullptr;   // chunk c runs on stream c mod (number of chunks a batch is split into): one chunk's narrow tree levels run under the others' wide ones
  DevGraph* d_gs = nullptr; size_t cap_gs = 0;
  BatchStage* d_stage = nullptr; size_t cap_stage = 0;
  BatchAlt* d_alt = nullptr; size_t cap_alt = 0;          // dual-lambda form: second factorisation + the three state copies per graph
  RestoreRec* restore_tab = nullptr; size_t cap_restore = 0;   // pinned: pps_multi_restore_state's (estimate, snapshot, length) per graph
  double* results = nullptr; size_t cap_results = 0;      // pinned: 12 doubles per graph (8 used by the single-lambda form)
  double seq = 0.0;
  int rounds = 0; double t_total = 0;
  // profiling (pps_multi_set_profiling): HIP events at the phase boundaries of every round, resolved after the solve
  int profiling = 0;
  std::vector<hipEvent_t> evs; size_t ev_used = 0;
  double t_phase[5] = {0, 0, 0, 0, 0};     // K1 | K2 | factor | back-substitution | trial step + chi2   [seconds, device]
  long long n_relin = 0, n_solves = 0;      // graphs re-linearised / factorised, summed over the rounds
};

static int mfail(pps_multi* m, int code, const std::string& msg) { if (m) m->err = msg; return code; }
#define MHIP(m, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return mfail(m, PPS_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

int pps_multi_create(int n, pps_graph* const* graphs, pps_multi** out) {
  if (!out || n < 1 || !graphs) return PPS_EINVAL;
  for (int i = 0; i < n; i++) {
    if (!graphs[i]) return PPS_EINVAL;
    if (graphs[i]->props.device != graphs[0]->props.device) return PPS_EINVAL;
    for (int j = 0; j < i; j++) if (graphs[j] == graphs[i]) return PPS_EINVAL;
  }
  pps_multi* m = new (std::nothrow) pps_multi();
  if (!m) return PPS_ENOMEM;
  m->gs.assign(graphs, graphs + n);
  m->device = graphs[0]->props.device;
  m->sw = read_switches();
  *out = m;
  return PPS_OK;
}

int pps_multi_destroy(pps_multi* m) {
  if (!m) return PPS_EINVAL;
  if (m->stream2) { (void)hipStreamSynchronize(m->stream2); (void)hipStreamDestroy(m->stream2); m->stream2 = nullptr; }
  if (m->stream3) { (void)hipStreamSynchronize(m->stream3); (void)hipStreamDestroy(m->stream3); m->stream3 = nullptr; }
  if (m->stream) {
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    (void)hipStreamDestroy(m->stream);
  }
  for (hipEvent_t e : m->evs) (void)hipEventDestroy(e);
  if (m->d_gs) (void)hipFree(m->d_gs);
  if (m->d_stage) (void)hipFree(m->d_stage);
  if (m->d_alt) (void)hipFree(m->d_alt);
  if (m->results) (void)hipHostFree(m->results);
  if (m->restore_tab) (void)hipHostFree(m->restore_tab);
  delete m;
  return PPS_OK;
}

const char* pps_multi_last_error(const pps_multi* m) { return m ? m->err.c_str() : "null handle"; }

static int multi_optimize(pps_multi* m, int* iterations, int* status);

int pps_multi_save_state(pps_multi* m) {
  if (!m) return PPS_EINVAL;
  m->err.clear();
  for (size_t i = 0; i < m->gs.size(); i++) {
    const int rc = pps_save_state(m->gs[i]);
    if (rc != PPS_OK) return mfail(m, rc, "graph " + std::to_string(i) + ": " + m->gs[i]->err);
  }
  for (pps_graph* g : m->gs) MHIP(m, hipStreamSynchronize(g->stream));
  return PPS_OK;
}

// pps_restore_state of every graph in ONE launch (128 graphs: 30 us instead of 128 copies on 128 streams and as many stream
// synchronisations at the next solve); complete on return
int pps_multi_restore_state(pps_multi* m) {
  if (!m) return PPS_EINVAL;
  m->err.clear();
  const int G = (int)m->gs.size();
  if (hipSetDevice(m->device) != hipSuccess) return mfail(m, PPS_EHIP, "hipSetDevice failed (no HIP device: there is no CPU fallback)");
  for (int i = 0; i < G; i++) {
    const pps_graph* g = m->gs[i];
    if (!g->snap_pose || g->snap_version != g->upload_version || g->topo_dirty) return mfail(m, PPS_ESTATE, "graph " + std::to_string(i) + ": no snapshot for the current topology");
    if (g->host_values_newer) return mfail(m, PPS_ESTATE, "graph " + std::to_string(i) + ": host values were modified after the snapshot");
  }
  if (m->cap_restore < (size_t)G) {
    if (m->restore_tab) (void)hipHostFree(m->restore_tab);
    m->restore_tab = nullptr; m->cap_restore = 0;
    MHIP(m, hipHostMalloc(reinterpret_cast<void**>(&m->restore_tab), sizeof(RestoreRec) * (size_t)G, hipHostMallocDefault));
    m->cap_restore = G;
  }
  if (!m->stream) MHIP(m, hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  for (int i = 0; i < G; i++) {
    pps_graph* g = m->gs[i];
    MHIP(m, hipStreamSynchronize(g->stream));                      // (idle unless the caller queued work on the handle itself)
    const DevGraph& d = g->dev;
    m->restore_tab[i] = RestoreRec{d.pose_est, g->snap_pose, (long long)((size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld)};
  }
  MHIP(m, launch_batch_restore(m->restore_tab, G, m->stream));
  MHIP(m, hipStreamSynchronize(m->stream));
  for (pps_graph* g : m->gs) { g->dev_values_newer = true; g->pin_holds_est = false; }
  return PPS_OK;
}

// The class lists of K2's throughput form (pps_device.h: k2t), once per upload of a graph that enters a large batch.
static int ensure_k2t_lists(pps_graph* g) {
  if (g->k2t_version == g->upload_version && g->dev.k2t) return PPS_OK;
  const Analysis& A = g->an;
  std::vector<int> big, small, generic;
  for (int sg : A.nd_segs) {
    const int* r = &A.srec[8 * (size_t)sg];
    const int rows = r[0], cols = r[1], size = r[2], c0 = r[3], cnt = r[4];
    int cls = 0;
    bool three = true, three_six = true, six = true, same = true;
    for (int k = c0; k < c0 + cnt; k++) {
      const int* c = &A.contrib[4 * (size_t)k];
      const bool prod = c[3] >= kProductFlag;
      const int mm = prod ? c[3] - kProductFlag : c[3];
      three = three && mm == 3; six = six && mm == 6; three_six = three_six && (mm == 3 || mm == 6);
      same = same && (prod || c[0] == c[1]);                    // row slice = column slice (a diagonal block)
    }
    if (cnt >= 1 && cnt <= 64 && rows == 3 && cols == 3 && size == 12 && three && same) { small.push_back(sg); continue; }
    if (cnt >= 1 && cnt <= 64 && rows == 6 && cols == 6 && size == 42 && three_six && same) cls = 1;
    else if (cnt >= 1 && cnt <= 64 && rows == 6 && cols == 6 && size == 36 && six) cls = 2;
    (cls ? big : generic).push_back(sg | (cls << 28));
  }
  // the entries with a class body first (kb_hblocks_tc: ~80 registers, six waves per SIMD), the generic ones behind them (kb_hblocks_tg)
  g->dev.n_k2t_spec = (int)big.size();
  big.insert(big.end(), generic.begin(), generic.end());
  g->dev.n_k2t_big = (int)big.size(); g->dev.n_k2t_small = (int)small.size();
  big.insert(big.end(), small.begin(), small.end());
  if (big.empty()) big.push_back(0);
  int rc = dev_alloc(g, &g->dev.k2t, big.size()); if (rc != PPS_OK) return rc;
  if (hipMemcpy(g->dev.k2t, big.data(), sizeof(int) * big.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(g, PPS_EHIP, "upload of the K2 class lists failed");
  g->k2t_version = g->upload_version;
  return PPS_OK;
}

int pps_multi_optimize(pps_multi* m, int* iterations, int* status) {
  if (!m) return PPS_EINVAL;
  const int rc = multi_optimize(m, iterations, status);
  if (rc != PPS_OK && rc != PPS_ENOTPD && rc != PPS_EINVAL && rc != PPS_ESTATE) {       // a HIP failure in the middle of the rounds: as a failed single solve
    // every stream a chunk may still be running on is drained BEFORE the handles are marked abandoned: the next upload_all frees or
    // re-uploads their arenas, and a kernel of this solve still in flight would write into them
    for (hipStream_t st : {m->stream, m->stream2, m->stream3}) if (st) (void)hipStreamSynchronize(st);
    for (pps_graph* g : m->gs) abandon_device_copy(g);
  }
  return rc;
}

static int multi_optimize(pps_multi* m, int* iterations, int* status) {
  const double t0 = now_s();
  const int G = (int)m->gs.size();
  if (hipSetDevice(m->device) != hipSuccess) return mfail(m, PPS_EHIP, "hipSetDevice failed (no HIP device: there is no CPU fallback)");
  // ---- every graph analysed, uploaded and idle; all of them must take the wave-per-front path ----
  int mode = m->gs[0]->props.jacobian_mode, max_stages = 0;
  for (int i = 0; i < G; i++) {
    pps_graph* g = m->gs[i];
    reset_solve_stats(g);
    g->tr_lambda.clear(); g->tr_chi2.clear(); g->tr_acc.clear();
    int rc = prepare_solve(g);
    if (rc != PPS_OK) return mfail(m, rc, "graph " + std::to_string(i) + ": " + g->err);
    if (!g->use_band) return mfail(m, PPS_ESTATE, "graph " + std::to_string(i) + " has fronts beyond the wave-per-front kernels (loop closures): solve it through its own handle");
    if (g->n_live_factors == 0) return mfail(m, PPS_ESTATE, "graph " + std::to_string(i) + " has no factors");
    if (g->props.jacobian_mode != mode) return mfail(m, PPS_EINVAL, "all graphs of a batch share one jacobian_mode");
    if (g->an.n_stages > 32) return mfail(m, PPS_ESTATE, "graph " + std::to_string(i) + ": elimination tree too deep for the batched schedule");
    rc = ensure_k2t_lists(g);
    if (rc != PPS_OK) return mfail(m, rc, "graph " + std::to_string(i) + ": " + g->err);
    if (hipStreamQuery(g->stream) != hipSuccess) MHIP(m, hipStreamSynchronize(g->stream));   // (work the caller queued on the handle itself)
    g->status_clean = false;
    max_stages = std::max(max_stages, g->an.n_stages);
  }
  const double t_s1 = now_s() - t0;
  // both damping values of a linearisation in the same launches (lm_solve_dual's scheme): every uploaded handle has its second
  // factor / state set
  for (int i = 0; i < G; i++)
    if (!(m->gs[i]->spec_L && m->gs[i]->spec_U && m->gs[i]->spec_delta && m->gs[i]->spec_pose && m->gs[i]->spec_result))
      return mfail(m, PPS_ESTATE, "graph " + std::to_string(i) + " has no second factor set (not uploaded)");
  const bool dual = true;
  if (!m->stream) MHIP(m, hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  if (!m->stream2) MHIP(m, hipStreamCreateWithFlags(&m->stream2, hipStreamNonBlocking));
  if (!m->stream3) MHIP(m, hipStreamCreateWithFlags(&m->stream3, hipStreamNonBlocking));
  if (m->cap_results < (size_t)G) {
    if (m->results) (void)hipHostFree(m->results);
    m->results = nullptr; m->cap_results = 0;
    MHIP(m, hipHostMalloc(reinterpret_cast<void**>(&m->results), sizeof(double) * 12 * (size_t)G, hipHostMallocDefault));
    m->cap_results = G;
  }
  memset(m->results, 0, sizeof(double) * 12 * (size_t)G);
  // ---- device tables: the graphs' records and their band schedules ----
  std::vector<DevGraph> hg(G);
  std::vector<BatchStage> hs((size_t)std::max(1, max_stages) * G, BatchStage{0, 0});
  for (int i = 0; i < G; i++) {
    hg[i] = m->gs[i]->dev;
    const Analysis& A = m->gs[i]->an;
    for (int stg = 0; stg < A.n_stages; stg++) hs[(size_t)stg * G + i] = BatchStage{A.stage_grp_off[stg], A.stage_grp_off[stg + 1] - A.stage_grp_off[stg]};
  }
  if (m->cap_gs < (size_t)G) { if (m->d_gs) (void)hipFree(m->d_gs); m->d_gs = nullptr; MHIP(m, hipMalloc(reinterpret_cast<void**>(&m->d_gs), sizeof(DevGraph) * (size_t)G)); m->cap_gs = G; }
  if (m->cap_stage < hs.size()) { if (m->d_stage) (void)hipFree(m->d_stage); m->d_stage = nullptr; MHIP(m, hipMalloc(reinterpret_cast<void**>(&m->d_stage), sizeof(BatchStage) * hs.size())); m->cap_stage = hs.size(); }
  MHIP(m, hipMemcpy(m->d_gs, hg.data(), sizeof(DevGraph) * (size_t)G, hipMemcpyHostToDevice));
  MHIP(m, hipMemcpy(m->d_stage, hs.data(), sizeof(BatchStage) * hs.size(), hipMemcpyHostToDevice));
  const double t_s2 = now_s() - t0;
  // ---- launch geometry per chunk of kBatchMax graphs ----
  int n_cu = 256;
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, m->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount; }
  // Large batches are split into two or three chunks of equal size that advance on their own streams: the launches of a chunk's upper tree
  // levels hold a few hundred wavefronts each and leave most of the device to the other chunks' kernels.  A chunk keeps at least 120 000
  // factors -- below that its kernels would be the latency forms.  Round 6 (with the throughput forms from 120 000 factors per chunk on, same
  // box, graphs/s): G = 40 as 1 chunk 1 724, as 2 x 20 1 809; G = 48 1 845 / 1 935; G = 64 1 950 / 2 091; G = 72 as 2 / 3 chunks 2 163 / 2 155;
  // G = 80 2 220 / 2 199; G = 96 2 234 / 2 269; G = 128 2 314 / 2 346 -- two chunks from 240 000 factors, three from 560 000.  (Event-timed
  // profiling keeps one stream and whole chunks: the phases must not overlap.)
  long long total_factors = 0;
  for (int i = 0; i < G; i++) total_factors += m->gs[i]->n_live_factors;
  const int n_split = m->sw.multi_split > 0 ? std::min(3, m->sw.multi_split)      // (PPS_MULTI_SPLIT: the multi-chunk scheduler on a small batch -- tests)
                                            : (total_factors >= 560000 ? 3 : (total_factors >= 240000 ? 2 : 1));
  const bool two_streams = n_split > 1 && !m->profiling;
  const int CH = two_streams ? std::min(kBatchMax, (G + n_split - 1) / n_split) : kBatchMax;
  const int n_chunks = (G + CH - 1) / CH;
  std::vector<BatchGeom> geom(n_chunks);
  const size_t lds_budget = 150 * 1024;
  for (int c = 0; c < n_chunks; c++) {
    BatchGeom& q = geom[c];
    q.n_stages = max_stages;
    int max_panel[32] = {0};
    for (int stg = 0; stg < 32; stg++) q.stage_reg_only[stg] = true;
    bool level_ok = true;
    bool lvl_direct_bad[64] = {false};
    for (int i = c * CH; i < std::min(G, (c + 1) * CH); i++) {
      const pps_graph* g = m->gs[i];
      const DevGraph& d = g->dev;
      const Analysis& A = g->an;
      q.lin_blocks = std::max(q.lin_blocks, (d.n_obs_fixed + 11) / 12 + (d.n_odo + 7) / 8 + (d.n_pp + 7) / 8 + (d.n_lp + 7) / 8);   // (lane form: 12 plane observations / 8 other factors per block)
      q.lin_obs_blocks = std::max(q.lin_obs_blocks, (d.n_obs_fixed + 127) / 128);
      q.lin_rest_blocks = std::max(q.lin_rest_blocks, (d.n_odo + 127) / 128 + (d.n_pp + 127) / 128 + (d.n_lp + 127) / 128);
      q.repop_blocks = std::max(q.repop_blocks, (d.n_obs - d.n_obs_fixed + 63) / 64);
      q.hblocks = std::max(q.hblocks, (d.n_nd_segs + 3) / 4);                     // (the segments K1 does not write itself)
      q.k2t_blocks = std::max(q.k2t_blocks, (d.n_k2t_spec + 15) / 16 + (d.n_k2t_small + 15) / 16);
      q.k2tg_blocks = std::max(q.k2tg_blocks, (d.n_k2t_big - d.n_k2t_spec + 15) / 16);
      q.hreduce = std::max(q.hreduce, d.n_mseg);
      q.k2_blocks = std::max(q.k2_blocks, (d.n_k2_single + 15) / 16 + d.n_k2_multi);
      q.k2_finish = std::max(q.k2_finish, d.n_k2_finish);
      q.retract = std::max(q.retract, (d.n_pose + d.n_plane + 255) / 256);
      q.chi2 = std::max(q.chi2, d.chi2_blocks);
      q.n_factors_total += (long long)d.n_obs + d.n_odo + d.n_pp + d.n_lp;
      q.n_levels = std::max(q.n_levels, A.n_levels);
      if (A.n_levels > 64 || A.max_front + 1 > band_reg_rows() || g->dev.trace) level_ok = false;
      for (int s2 = 0; s2 < A.n_fronts; s2++) {
        const int l = A.f_level[s2];
        if (l >= 0 && l < 64) {
          q.lvl_max_panel[l] = std::max(q.lvl_max_panel[l], (A.f_p[s2] + A.f_b[s2] + 1) * A.f_p[s2]);
          if (!band_level_solve_direct_ok(A.f_p[s2], A.f_b[s2])) lvl_direct_bad[l] = true;
          q.lvl_direct_pp[l] = std::max(q.lvl_direct_pp[l], A.f_p[s2] * A.f_p[s2]);
        }
      }
      for (int l = 0; l < A.n_levels && l < 64; l++) {
        for (int c2 = 0; c2 < 3; c2++) q.lvl_cls_blocks[l][c2] = std::max(q.lvl_cls_blocks[l][c2], (A.cls_off[3 * l + c2 + 1] - A.cls_off[3 * l + c2] + 3) / 4);
        q.lvl_blocks[l] = std::max(q.lvl_blocks[l], (A.cls_off[3 * l + 3] - A.cls_off[3 * l] + 3) / 4);
      }
      for (int stg = 0; stg < A.n_stages; stg++) {
        q.stage_groups[stg] = std::max(q.stage_groups[stg], A.stage_grp_off[stg + 1] - A.stage_grp_off[stg]);
        q.stage_nw_factor[stg] = std::max(q.stage_nw_factor[stg], g->stage_nw_factor[stg]);
        q.stage_nw_solve[stg] = std::max(q.stage_nw_solve[stg], g->stage_nw_solve[stg]);
        q.stage_per_wave_factor[stg] = std::max(q.stage_per_wave_factor[stg], A.stage_max_front[stg]);   // (max front for now: sized below)
        q.stage_max_front[stg] = std::max(q.stage_max_front[stg], A.stage_max_front[stg]);
        max_panel[stg] = std::max(max_panel[stg], g->stage_max_panel[stg]);
        q.stage_grp_fronts[stg] = std::max(q.stage_grp_fronts[stg], g->stage_max_grp_fronts[stg]);
        if (A.stage_max_front[stg] + 1 > band_reg_rows() || g->dev.trace) q.stage_reg_only[stg] = false;
      }
    }
    for (int l = 0; l < 64; l++) if (lvl_direct_bad[l]) q.lvl_direct_pp[l] = 0;
    // the lane-parallel central differences (32 lanes per factor, 13 of them idle) are the low-latency form; from a few
    // hundred thousand factors per launch the thread-per-factor form has the higher throughput
    const long long thr = m->sw.multi_thread_factors;          // 120 000 (PPS_MULTI_THREAD_FACTORS lowers it for the tests)
    q.lin_thread_form = q.n_factors_total > thr || m->sw.multi_thread_form;      // (PPS_MULTI_THREAD_FORM / PPS_MULTI_LEVELS: forced onto small batches by the parity test)
    // throughput over latency from the same size on: a launch per tree level and size class instead of a launch per band
    q.level_form = level_ok && (q.n_factors_total > thr || m->sw.multi_levels);
    { int mp = 1; for (int stg = 0; stg < max_stages; stg++) mp = std::max(mp, max_panel[stg]); q.solve_per_wave_all = (int)(band_solve_lds_bytes(mp) / sizeof(double)); }
    for (int stg = 0; stg < max_stages; stg++) {
      q.stage_per_wave_factor[stg] = (int)(band_lds_bytes(q.stage_per_wave_factor[stg], q.stage_reg_only[stg]) / sizeof(double));
      q.stage_per_wave_solve[stg] = (int)(band_solve_lds_bytes(max_panel[stg]) / sizeof(double));
      const size_t fw = (size_t)q.stage_per_wave_factor[stg] * sizeof(double), sw = (size_t)q.stage_per_wave_solve[stg] * sizeof(double);
      const size_t xbytes = (size_t)q.stage_grp_fronts[stg] * band_max_rows() * sizeof(double);
      if (fw > lds_budget || xbytes + sw > lds_budget) return mfail(m, PPS_ESTATE, "a band group of this batch does not fit the LDS: solve the graphs through their own handles");
      q.stage_nw_factor[stg] = (int)std::max<size_t>(1, std::min<size_t>(q.stage_nw_factor[stg], lds_budget / fw));
      q.stage_nw_solve[stg] = (int)std::max<size_t>(1, std::min<size_t>(q.stage_nw_solve[stg], (lds_budget - xbytes) / sw));
      // Throughput, not latency, is what a batch is for.  A band group is a sub-tree (8 + 4 + 2 + 1 fronts on C2): walked
      // by 8 waves, half of the wave-slots -- and the LDS they hold -- idle on its upper levels.  When the chunk has more
      // groups than the device has wave-slots, fewer waves per group keep every slot on a front (2 waves: 94 % instead of
      // 47 %); the groups of the upper stages stay wide, there the tree depth is the cost.
      long long total_groups = 0;
      for (int i = c * CH; i < std::min(G, (c + 1) * CH); i++) {
        const Analysis& A = m->gs[i]->an;
        if (stg < A.n_stages) total_groups += (dual ? 2 : 1) * (A.stage_grp_off[stg + 1] - A.stage_grp_off[stg]);
      }
      if (total_groups > 0) {
        // (wave-slots of a CU: what its LDS holds, and no more than the registers allow -- 2 waves per SIMD for the factor kernels, 3 for the
        // back-substitution: G = 8 is 512 groups of stage 0, and with eight waves each only 256 of them were resident at a time)
        // (counted by the LDS alone, up to round 4: G = 8 786 graphs/s against 865, G = 16 1 138 against 1 176, G = 24 1 222 against 1 259)
        const long long slots_f = (long long)n_cu * std::min<size_t>(8, std::max<size_t>(1, lds_budget / fw));
        const long long slots_s = (long long)n_cu * std::min<size_t>(12, std::max<size_t>(1, (lds_budget - std::min(lds_budget / 2, xbytes)) / sw));
        q.stage_nw_factor[stg] = (int)std::max<long long>(1, std::min<long long>(q.stage_nw_factor[stg], (slots_f + total_groups - 1) / total_groups));
        q.stage_nw_solve[stg] = (int)std::max<long long>(1, std::min<long long>(q.stage_nw_solve[stg], (slots_s + total_groups - 1) / total_groups));
      }
    }
  }
  // the pre-assembling walk (k_band_factor_pre) and the data-flow back-substitution (k_band_solve_flow) of the band kernels, per chunk
  for (int c = 0; c < n_chunks; c++) {
    BatchGeom& q = geom[c];
    for (int stg = 0; stg < q.n_stages && stg < 32; stg++) {
      bool pre = q.stage_reg_only[stg] && !m->sw.no_preassemble;
      const int nw = q.stage_nw_factor[stg];
      for (int i = c * CH; pre && i < std::min(G, (c + 1) * CH); i++) {
        const Analysis& A = m->gs[i]->an;
        if (stg >= A.n_stages) continue;
        for (int gi = A.stage_grp_off[stg]; pre && gi < A.stage_grp_off[stg + 1]; gi++) {
          const int l0 = A.grp_lvl_off[gi], nl = A.grp_lvl_off[gi + 1] - l0;
          pre = nl >= 1 && nl <= 4;
          int upper = 0, c0 = 0;
          for (int k = 0; pre && k < nl; k++) {
            const int cnt = A.glvl_front_off[l0 + k + 1] - A.glvl_front_off[l0 + k];
            if (k == 0) c0 = cnt; else upper += cnt;
          }
          pre = pre && (c0 + upper <= nw || (nl >= 3 && c0 <= nw && upper <= nw));      // (shape B | shape A of body_band_factor_pre)
        }
      }
      q.stage_pre[stg] = pre;
      // data flow: as many waves as the largest group has fronts, at most twelve and what the LDS holds.  A stage whose groups were
      // narrowed to fill the device with groups (throughput: G = 32 and up) keeps the barrier form -- waves that spin on a flag hold
      // wave slots other groups could use (G = 32: 3.9 against 3.7 ms of back-substitution per batch solve)
      q.stage_nw_flow[stg] = 0;
      if (!m->sw.no_solve_flow && q.stage_grp_fronts[stg] > 1 && q.stage_nw_solve[stg] >= std::min(8, q.stage_grp_fronts[stg])) {
        const size_t per_wave = (size_t)q.stage_per_wave_solve[stg] * sizeof(double);
        const size_t fixed = ((size_t)q.stage_grp_fronts[stg] * band_max_rows() + (size_t)(q.stage_grp_fronts[stg] + 1) / 2) * sizeof(double);
        const size_t lds_budget = 150 * 1024;
        if (fixed + per_wave <= lds_budget) {
          const int room = (int)((lds_budget - fixed) / per_wave);
          int nwf = std::min(std::min(12, q.stage_grp_fronts[stg]), room);
          q.stage_nw_flow[stg] = std::max(1, nwf);
        }
      }
    }
  }
  // ---- dual-lambda form: every graph walks lm_solve_dual's scheme, in lockstep rounds of one linearisation each ----
  const double t_setup = now_s() - t0;
  const bool timing_rounds = m->sw.multi_timing > 1;
  m->n_chunks_last = n_chunks; m->forms_last = 0;
  for (int c = 0; c < n_chunks; c++) m->forms_last |= (geom[c].lin_thread_form ? 1 : 0) | (geom[c].level_form ? 2 : 0);

  {
    std::vector<BatchAlt> ha(G);
    for (int i = 0; i < G; i++) {
      pps_graph* g = m->gs[i];
      ha[i] = BatchAlt{g->spec_L, g->spec_U, g->spec_delta, g->spec_result, g->spec_chi2_partials, g->spec_dn_partials, g->spec_ticket,
                       {g->dev.pose_est, g->dev.pose_lin, g->spec_pose}, {g->dev.plane_est, g->dev.plane_lin, g->spec_plane}};
    }
    if (m->cap_alt < (size_t)G) { if (m->d_alt) (void)hipFree(m->d_alt); m->d_alt = nullptr; MHIP(m, hipMalloc(reinterpret_cast<void**>(&m->d_alt), sizeof(BatchAlt) * (size_t)G)); m->cap_alt = G; }
    MHIP(m, hipMemcpy(m->d_alt, ha.data(), sizeof(BatchAlt) * (size_t)G, hipMemcpyHostToDevice));
    struct LMD { double lambda, error, dnorm; int num_iter, cur, xsel; bool done, have_next, relin, active, last_notpd, trial_taken; int n_notpd; };
    std::vector<LMD> lm(G);
    for (int i = 0; i < G; i++) lm[i] = LMD{m->gs[i]->props.lm_lambda0, 0.0, 0.0, 0, 0, 0, false, true, true, true, false, false, 0};
    auto make_args = [&](int c, double seq) {
      BatchArgs a{};
      a.gs = m->d_gs; a.stage_tab = m->d_stage; a.results = m->results; a.n_total = G; a.b0 = c * CH;
      a.n = std::min(G, (c + 1) * CH) - a.b0; a.seq = seq;
      a.alt = m->d_alt; a.rstride = 12;
      a.no_products = geom[c].lin_thread_form ? 1 : 0;
      for (int k = 0; k < a.n; k++) {
        const LMD& q = lm[a.b0 + k];
        a.lambda[k] = q.lambda; a.lambda2[k] = q.lambda * m->gs[a.b0 + k]->props.lm_lambda_factor;
        a.xsel[k] = (unsigned char)q.xsel;
        a.flags[k] = (unsigned char)((q.active ? BF_ACTIVE : 0) | (q.relin ? BF_RELIN : 0));
      }
      return a;
    };
    m->ev_used = 0; m->n_relin = 0; m->n_solves = 0;
    for (double& t : m->t_phase) t = 0;
    auto mark = [&]() {
      if (!m->profiling) return;
      if (m->ev_used == m->evs.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; m->evs.push_back(e); }
      (void)hipEventRecord(m->evs[m->ev_used++], m->stream);
    };
    auto next_event = [&]() -> hipEvent_t {
      if (!m->profiling) return nullptr;
      if (m->ev_used == m->evs.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; m->evs.push_back(e); }
      return m->evs[m->ev_used++];
    };
    // one round of one chunk: e0 | K1 | e1 | K2 (+ chi2 at x) | e2 | factor x 2 | e3 | solve x 2 | e4 | both trials | e5
    auto run_round = [&](const BatchArgs& a, const BatchGeom& q, bool first, bool any_relin, hipStream_t st) -> int {
      if (first) MHIP(m, launch_batch_begin_dual(a, q, st));
      mark();
      if (any_relin) MHIP(m, launch_batch_linearize(a, q, q.lin_thread_form ? mode | 2 : mode, st));
      mark();
      if (any_relin) MHIP(m, launch_batch_hblocks(a, q, st, !q.lin_thread_form));
      if (first) MHIP(m, launch_batch_chi2(a, q, 0, st));
      mark();
      hipEvent_t ef = next_event();
      MHIP(m, launch_batch_solve(a, q, st, ef));
      mark();
      MHIP(m, launch_batch_trial_dual(a, q, st));
      mark();
      return PPS_OK;
    };
    // the part of lm_solve_dual's loop that needs no launch: consume the verdicts that are on the host.  Returns with the
    // graph done, or active (and possibly relin) for the next round.
    auto advance = [&](int i) {
      LMD& q = lm[i];
      pps_graph* g = m->gs[i];
      const pps_props& prop = g->props;
      q.active = false; q.relin = false;
      for (;;) {
        if (!((prop.max_iterations <= 0 || q.num_iter < prop.max_iterations) && q.dnorm > prop.epsilon2 && q.error > prop.epsilon_abs)) { q.done = true; return; }
        q.num_iter++;
        const double* rec = m->results + 12 * (size_t)i + 4 * (1 + q.cur);
        const double error_new = rec[0];
        const double error_diff = q.error - error_new;
        const bool accepted = error_diff > 0.;
        g->tr_lambda.push_back(q.lambda); g->tr_chi2.push_back(error_new); g->tr_acc.push_back(accepted ? 1 : 0);
        if (accepted) {
          g->stats.lm_trials_accepted++;
          if (error_diff < prop.epsilon_rel * q.error) { q.error = error_new; q.trial_taken = true; q.done = true; return; }   // (:431-434)
          q.lambda /= prop.lm_lambda_factor;
          q.error = error_new;
          q.xsel = (q.xsel + 1 + q.cur) % 3;                           // the accepted copy is the linearisation point now
          q.relin = true; q.active = true; q.cur = 0; q.have_next = true;
          g->stats.n_linearize++; g->stats.n_factorize += 2;
          return;
        }
        g->stats.lm_trials_rejected++;
        q.lambda *= prop.lm_lambda_factor;
        if (q.have_next) {                                             // the step for this lambda was computed alongside
          q.cur = 1; q.have_next = false;
          const double* rb = m->results + 12 * (size_t)i + 8;
          q.dnorm = std::sqrt(rb[1]); q.last_notpd = rb[2] != 0.0; q.n_notpd += q.last_notpd ? 1 : 0;
          continue;
        }
        q.active = true; q.cur = 0; q.have_next = true;               // both rejected: same J and H, two more damping values
        g->stats.n_factorize += 2;
        return;
      }
    };
    // The chunks advance on their own: a chunk's next round is launched as soon as ITS graphs have delivered their records, while the
    // other chunk's kernels keep the device busy -- no barrier over the whole batch between rounds (the device would idle for the
    // host's bookkeeping 42 times per solve), and the chunks drift apart, so that one's narrow tree levels meet the other's wide ones.
    struct ChunkRun { double seq = 0.0; bool in_flight = false, first = true; int rounds = 0; };
    std::vector<ChunkRun> cr(n_chunks);
    hipStream_t const streams[3] = {m->stream, m->stream2, m->stream3};
    auto stream_of = [&](int c) { return n_chunks > 1 && !m->profiling ? streams[c % n_split] : m->stream; };
    auto chunk_done = [&](int c) -> bool {                     // (non-blocking) both records of every active graph of the chunk are this round's
      for (int i = c * CH; i < std::min(G, (c + 1) * CH); i++) {
        if (!lm[i].active) continue;
        const volatile double* r = m->results + 12 * (size_t)i;
        if (r[4 + 3] != cr[c].seq || r[8 + 3] != cr[c].seq) return false;
      }
      return true;
    };
    m->rounds = 0;
    for (int c = 0; c < n_chunks; c++) {
      m->seq += 1.0; cr[c].seq = m->seq; cr[c].in_flight = true;
      const BatchArgs a = make_args(c, cr[c].seq);
      int rc = run_round(a, geom[c], true, true, stream_of(c)); if (rc != PPS_OK) return rc;
    }
    m->n_relin += G; m->n_solves += 2 * (long long)G;
    int n_flight = n_chunks;
    double t_progress = now_s();
    // a HIP failure inside the loop: whatever the other chunks still have in flight is drained before the caller sees the error
    auto drain = [&]() { for (hipStream_t s3 : streams) if (s3) (void)hipStreamSynchronize(s3); };
    double t_launch = 0.0, t_book = 0.0;                       // host seconds inside run_round / between a chunk's records and its next launch
    int n_launch_rounds = 0;
    while (n_flight > 0) {
      bool progressed = false;
      for (int c = 0; c < n_chunks; c++) {
        if (!cr[c].in_flight || !chunk_done(c)) continue;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const double tb0 = m->sw.multi_timing ? now_s() : 0.0;
        progressed = true;
        cr[c].rounds++;
        const int i0 = c * CH, i1 = std::min(G, (c + 1) * CH);
        int n_active = 0;
        for (int i = i0; i < i1; i++) {
          pps_graph* g = m->gs[i];
          if (cr[c].first) {
            const double* r0 = m->results + 12 * (size_t)i;
            lm[i].error = r0[0]; g->stats.chi2_initial = r0[0];
            lm[i].dnorm = std::sqrt(r0[5]); lm[i].last_notpd = r0[6] != 0.0; lm[i].n_notpd = lm[i].last_notpd ? 1 : 0;
            g->stats.n_linearize = 1; g->stats.n_factorize = 2;
          } else if (lm[i].active) {
            const double* r1 = m->results + 12 * (size_t)i + 4;
            lm[i].dnorm = std::sqrt(r1[1]);
            lm[i].last_notpd = r1[2] != 0.0;
            lm[i].n_notpd += lm[i].last_notpd ? 1 : 0;
          }
          if (lm[i].active || cr[c].first) {                      // status words of this round's records (see wait_result, pps_solve.cpp)
            const double* rr = m->results + 12 * (size_t)i;
            if (rr[4 + 2] >= kStatusInternal || rr[8 + 2] >= kStatusInternal) {
              drain();
              return mfail(m, PPS_EHIP, "graph " + std::to_string(i) + ": internal error: a hand-over flag between the waves or workgroups of a K3 launch never arrived");
            }
          }
          if (!lm[i].done) advance(i); else { lm[i].active = false; lm[i].relin = false; }
          n_active += lm[i].active ? 1 : 0;
        }
        if (timing_rounds) fprintf(stderr, "  chunk %d round %d at %.3f ms: %d active next\n", c, cr[c].rounds, 1e3 * (now_s() - t0), n_active);
        cr[c].first = false;
        if (n_active == 0) { cr[c].in_flight = false; n_flight--; continue; }
        m->seq += 1.0; cr[c].seq = m->seq;
        const BatchArgs a = make_args(c, cr[c].seq);
        bool any_relin = false;
        for (int k = 0; k < a.n; k++) { any_relin = any_relin || (a.flags[k] & BF_RELIN); m->n_solves += (a.flags[k] & BF_ACTIVE) ? 2 : 0; m->n_relin += (a.flags[k] & BF_RELIN) ? 1 : 0; }
        const double tb1 = m->sw.multi_timing ? now_s() : 0.0;
        int rc = run_round(a, geom[c], false, any_relin, stream_of(c)); if (rc != PPS_OK) { drain(); return rc; }
        if (m->sw.multi_timing) { t_book += tb1 - tb0; t_launch += now_s() - tb1; n_launch_rounds++; }
      }
      if (progressed) { t_progress = now_s(); continue; }
      if (now_s() - t_progress > 2.0) {                        // (nothing for two seconds: let the streams drain, look once more)
        MHIP(m, hipStreamSynchronize(m->stream));
        MHIP(m, hipStreamSynchronize(m->stream2));
        MHIP(m, hipStreamSynchronize(m->stream3));
        bool any_done = false;
        for (int c = 0; c < n_chunks; c++) any_done = any_done || (cr[c].in_flight && chunk_done(c));
        if (!any_done) return mfail(m, PPS_EHIP, "result records of a round did not arrive");     // (all three streams are idle here)
      }
    }
    for (int c = 0; c < n_chunks; c++) m->rounds = std::max(m->rounds, cr[c].rounds);
    MHIP(m, hipStreamSynchronize(m->stream));
    MHIP(m, hipStreamSynchronize(m->stream2));
    MHIP(m, hipStreamSynchronize(m->stream3));
    for (size_t k = 0; k + 6 <= m->ev_used; k += 6) {
      const hipEvent_t* e = &m->evs[k];
      for (int ph = 0; ph < 5; ph++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e[ph], e[ph + 1]) == hipSuccess) m->t_phase[ph] += 1e-3 * ms;
      }
    }
    int first_bad = PPS_OK;
    m->t_total = now_s() - t0;
    if (m->sw.multi_timing)
      fprintf(stderr, "pps_multi: G %d total %.3f ms, of which setup %.3f (per-graph checks %.3f, tables %.3f, geometry %.3f); %d rounds\n", G,
              1e3 * m->t_total, 1e3 * t_setup, 1e3 * t_s1, 1e3 * (t_s2 - t_s1), 1e3 * (t_setup - t_s2), m->rounds);
    if (m->sw.multi_timing && n_launch_rounds > 0)
      fprintf(stderr, "pps_multi: host side of %d chunk rounds: launches %.3f ms (%.1f us per round), verdicts + arguments %.3f ms (%.1f us per round)\n",
              n_launch_rounds, 1e3 * t_launch, 1e6 * t_launch / n_launch_rounds, 1e3 * t_book, 1e6 * t_book / n_launch_rounds);
    for (int i = 0; i < G; i++) {
      pps_graph* g = m->gs[i];
      const LMD& q = lm[i];
      // linpoint_to_estimate (:466): the accepted, converged trial -- or the linearisation point when the pending step is dropped
      const int fin = q.trial_taken ? (q.xsel + 1 + q.cur) % 3 : q.xsel;
      double* const sp[3] = {ha[i].pose[0], ha[i].pose[1], ha[i].pose[2]};
      double* const sl[3] = {ha[i].plane[0], ha[i].plane[1], ha[i].plane[2]};
      g->dev.pose_est = sp[fin]; g->dev.plane_est = sl[fin];
      g->dev.pose_lin = sp[(fin + 1) % 3]; g->dev.plane_lin = sl[(fin + 1) % 3];
      g->spec_pose = sp[(fin + 2) % 3]; g->spec_plane = sl[(fin + 2) % 3];
      g->dev_values_newer = true; g->pin_holds_est = false;
      g->stats.lm_iterations = q.num_iter; g->stats.chi2_final = q.error; g->stats.lambda_final = q.lambda; g->stats.last_delta_norm = q.dnorm;
      g->stats.lm_trials_notpd = q.n_notpd; g->stats.t_total = m->t_total;
      if (iterations) iterations[i] = q.num_iter;
      const int st_i = q.last_notpd ? PPS_ENOTPD : PPS_OK;
      if (st_i != PPS_OK) g->err = "normal equations not positive definite at the last LM trial";
      if (status) status[i] = st_i;
      if (st_i != PPS_OK && first_bad == PPS_OK) first_bad = st_i;
    }
    if (first_bad != PPS_OK) return mfail(m, first_bad, "at least one graph ended on a factorisation that was not positive definite (see status[])");
    return PPS_OK;
  }
}

int pps_multi_set_profiling(pps_multi* m, int level) { if (!m) return PPS_EINVAL; m->profiling = level > 0 ? 1 : 0; return PPS_OK; }

int pps_multi_phase_times(const pps_multi* m, double sec[5], long long counts[4]) {
  if (!m || !sec) return PPS_EINVAL;
  for (int k = 0; k < 5; k++) sec[k] = m->t_phase[k];
  if (counts) { counts[0] = m->n_relin; counts[1] = m->n_solves; counts[2] = m->n_chunks_last; counts[3] = m->forms_last; }
  return PPS_OK;
}

int pps_multi_rounds(const pps_multi* m, int* rounds) { if (!m || !rounds) return PPS_EINVAL; *rounds = m->rounds; return PPS_OK; }

}  // extern "C"
