// pps_api.cpp -- C-ABI implementation, graph bookkeeping: handles and properties, nodes and factors, node values, state
// snapshots, introspection (stats, LM trace, per-factor r / J, analysis dump) and the K1 micro-benchmarks.  pps_graph.h lists the
// other implementation files.
#include "pps_graph.h"

using namespace pps;
using namespace pps_impl;

namespace pps {
// The library's only reader of the environment: once per handle (see Switches, pps_device.h).
Switches read_switches() {
  auto on = [](const char* k) { return getenv(k) != nullptr; };
  auto num = [](const char* k, long long dflt) { const char* e = getenv(k); return e && *e ? atoll(e) : dflt; };
  Switches s;
  s.k1_thread_form = on("PPS_K1_THREAD_FORM");
  {
    long long plain = num("PPS_PLAIN_SCHEDULE", 0);
    if (on("PPS_PLAIN_SCHEDULE") && plain <= 0) plain = 15;
    s.no_preassemble = (plain & 1) != 0; s.no_solve_flow = (plain & 2) != 0; s.no_root_fuse = (plain & 4) != 0; s.split_expand = (plain & 8) != 0;
  }
  s.no_spec_lin = on("PPS_NO_SPEC_LIN");
  s.no_dual = on("PPS_NO_DUAL");
  s.no_strip = on("PPS_NO_STRIP");
  s.no_incremental = on("PPS_NO_INCREMENTAL");
  s.no_incr_compact = on("PPS_NO_INCR_COMPACT");
  s.verify_upload = on("PPS_DEBUG_VERIFY_UPLOAD");
  s.multi_levels = on("PPS_MULTI_LEVELS");
  s.multi_thread_form = on("PPS_MULTI_THREAD_FORM");
  s.debug_drop_flag = (int)num("PPS_DEBUG_DROP_FLAG", 0);
  if (on("PPS_DEBUG_DROP_FLAG") && s.debug_drop_flag < 1) s.debug_drop_flag = 1;
  s.trace = (int)num("PPS_TRACE", 0);
  if (on("PPS_TRACE") && s.trace < 1) s.trace = 1;
  {
    long long tm = num("PPS_TIMING", 0);
    if (on("PPS_TIMING") && tm <= 0) tm = 7;
    s.analysis_timing = (tm & 1) != 0; s.upload_timing = (tm & 2) != 0; s.multi_timing = (tm & 8) ? 2 : ((tm & 4) ? 1 : 0);
  }
  s.multi_split = (int)num("PPS_MULTI_SPLIT", 0);
  s.multi_thread_factors = num("PPS_MULTI_THREAD_FACTORS", 120000);
  return s;
}
}  // namespace pps

extern "C" {

void pps_default_props(pps_props* p) {
  if (!p) return;
  p->epsilon2 = 1e-2 * 0.1;      // Properties.h:94 x Mapping.cpp:37
  p->epsilon_abs = 1e-3 * 0.1;   // Properties.h:98 x Mapping.cpp:38
  p->epsilon_rel = 1e-5 * 0.1;   // Properties.h:99 x Mapping.cpp:39
  p->max_iterations = 500;
  p->lm_lambda0 = 1e-6;
  p->lm_lambda_factor = 10.;
  p->jacobian_mode = PPS_JAC_NUMERIC;
  p->device = 0;
  p->verbose = 0;
}

int pps_version(void) { return PPS_VERSION; }

const char* pps_last_error(const pps_graph* g) { return g ? g->err.c_str() : "null handle"; }

int pps_graph_create(const pps_props* props, pps_graph** out) {
  if (!out) return PPS_EINVAL;
  pps_graph* g = new (std::nothrow) pps_graph();
  if (!g) return PPS_ENOMEM;
  if (props) g->props = *props; else pps_default_props(&g->props);
  g->sw = read_switches();
  g->aprm.timing = g->sw.analysis_timing ? 1 : 0;
  *out = g;
  return PPS_OK;
}

int pps_graph_destroy(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  if (!g->up_laps.empty()) {
    std::vector<std::pair<std::string, double>> v(g->up_laps.begin(), g->up_laps.end());
    std::sort(v.begin(), v.end());
    for (auto& kv : v) fprintf(stderr, "[upload] %-34s %9.3f ms total\n", kv.first.c_str(), 1e3 * kv.second);
  }
  if (g->acache) analysis_cache_free(g->acache);
  if (g->dev_ready) {
    (void)hipSetDevice(g->props.device);
    (void)hipStreamSynchronize(g->stream);
    free_device(g);
    release_arenas(g);
    if (g->host_result) (void)hipHostFree(g->host_result);
    if (g->ev[0]) (void)hipEventDestroy(g->ev[0]);
    if (g->ev[1]) (void)hipEventDestroy(g->ev[1]);
    for (hipEvent_t e : g->k1_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : g->fk_events) (void)hipEventDestroy(e);
    if (g->d_lms) (void)hipFree(g->d_lms);
    if (g->h_lms) (void)hipHostFree(g->h_lms);
    if (g->d_queries) (void)hipHostFree(g->d_queries);
    if (g->d_results) (void)hipHostFree(g->d_results);
    if (g->d_lm_planes) (void)hipFree(g->d_lm_planes);
    if (g->rp_pin) (void)hipHostFree(g->rp_pin);
    (void)hipStreamDestroy(g->stream);
  }
  delete g;
  return PPS_OK;
}

int pps_get_props(const pps_graph* g, pps_props* out) {
  if (!g || !out) return PPS_EINVAL;
  *out = g->props;
  return PPS_OK;
}
int pps_set_props(pps_graph* g, const pps_props* p) {
  if (!g || !p) return PPS_EINVAL;
  if (g->dev_ready && p->device != g->props.device) return fail(g, PPS_ESTATE, "device cannot change after the first solve");
  g->props = *p;
  return PPS_OK;
}

static int add_node(pps_graph* g, int type, const double* v, int nv, int* id) {
  if (!g || !v) return PPS_EINVAL;
  for (int k = 0; k < nv; k++) if (!std::isfinite(v[k])) return fail(g, PPS_EINVAL, "non-finite node value");
  if (g->dev_values_newer) { int rc = download_state(g); if (rc != PPS_OK) return rc; }
  HostNode n{};
  n.type = type;
  for (int k = 0; k < nv; k++) n.v[k] = v[k];
  if (type == NODE_PLANE) normalize4(n.v);
  n.deleted = false; n.compact = n.slot = -1;
  g->nodes.push_back(n);
  g->n_live_nodes++;
  g->dim_nodes += type == NODE_POSE ? 6 : 3;
  g->topo_dirty = true; g->analysis_stale = true; g->host_values_newer = true;
  if (id) *id = (int)g->nodes.size() - 1;
  return PPS_OK;
}

int pps_add_pose(pps_graph* g, const double tq[7], int* id) { return add_node(g, NODE_POSE, tq, 7, id); }
int pps_add_plane(pps_graph* g, const double abcd[4], int* id) { return add_node(g, NODE_PLANE, abcd, 4, id); }

static int add_factor(pps_graph* g, int type, int a, int b, const double* meas, int nm, const double* ut, int nw, int* fid) {
  if (!g || !meas || !ut) return PPS_EINVAL;
  for (int k = 0; k < nm; k++) if (!std::isfinite(meas[k])) return fail(g, PPS_EINVAL, "non-finite measurement");
  for (int k = 0; k < nw; k++) if (!std::isfinite(ut[k])) return fail(g, PPS_EINVAL, "non-finite sqrtinf");
  HostFactor f{};
  f.type = type; f.a = a; f.b = b; f.deleted = false; f.slot = -1;
  for (int k = 0; k < nm; k++) f.meas[k] = meas[k];
  if (nm == 4) normalize4(f.meas);
  for (int k = 0; k < nw; k++) f.w[k] = ut[k];
  g->factors.push_back(f);
  g->n_live_factors++;
  g->n_live_type[type]++;
  g->dim_measure += kFDim[type];
  g->topo_dirty = true; g->analysis_stale = true;
  if (fid) *fid = (int)g->factors.size() - 1;
  return PPS_OK;
}

int pps_add_pose_prior(pps_graph* g, int pose, const double meas6[6], const double ut[21], int* fid) {
  if (!g) return PPS_EINVAL;
  if (!live_node(g, pose, NODE_POSE)) return fail(g, PPS_EINVAL, "pose prior: unknown pose id");
  return add_factor(g, F_POSE_PRIOR, pose, -1, meas6, 6, ut, 21, fid);
}
int pps_add_odometry(pps_graph* g, int p1, int p2, const double meas6[6], const double ut[21], int* fid) {
  if (!g) return PPS_EINVAL;
  if (!live_node(g, p1, NODE_POSE) || !live_node(g, p2, NODE_POSE) || p1 == p2) return fail(g, PPS_EINVAL, "odometry: bad pose ids");
  return add_factor(g, F_ODOMETRY, p1, p2, meas6, 6, ut, 21, fid);
}
int pps_add_plane_obs(pps_graph* g, int pose, int plane, const double meas4[4], const double ut[6], int* fid) {
  if (!g) return PPS_EINVAL;
  if (!live_node(g, pose, NODE_POSE) || !live_node(g, plane, NODE_PLANE)) return fail(g, PPS_EINVAL, "plane obs: bad node ids");
  return add_factor(g, F_PLANE_OBS, pose, plane, meas4, 4, ut, 6, fid);
}
// Pose3d_Plane3d_Factor2 (src/isam_plane3d.h:314-424): same nodes / noise / log-map residual, but the measured
// plane is re-derived from the edge's two ground rays and the CURRENT pose at every evaluation.
int pps_add_plane_obs2(pps_graph* g, int pose, int plane, const double meas4[4], const double ray6[6], const double ut[6],
                       int* fid) {
  if (!g || !ray6) return PPS_EINVAL;
  for (int k = 0; k < 6; k++) if (!std::isfinite(ray6[k])) return fail(g, PPS_EINVAL, "non-finite edge ray");
  int id = -1;
  int rc = pps_add_plane_obs(g, pose, plane, meas4, ut, &id);
  if (rc != PPS_OK) return rc;
  g->factors[id].repop = 1;
  g->n_live_repop++;
  memcpy(g->factors[id].ray, ray6, sizeof g->factors[id].ray);
  if (fid) *fid = id;
  return PPS_OK;
}

// precompute_edge_ray (src/isam_plane3d.h:361-373): fp32 product invK * (u,v,1) per end point, cast to fp64
int pps_edge_ray(const float invK[9], const float seg2d[4], double ray6[6]) {
  if (!invK || !seg2d || !ray6) return PPS_EINVAL;
  for (int e = 0; e < 2; e++) {
    const float u = seg2d[2 * e], v = seg2d[2 * e + 1];
    for (int i = 0; i < 3; i++) ray6[3 * e + i] = (double)(invK[i * 3 + 0] * u + invK[i * 3 + 1] * v + invK[i * 3 + 2] * 1.f);
  }
  return PPS_OK;
}

int pps_add_plane_prior(pps_graph* g, int plane, const double meas4[4], const double ut[6], int* fid) {
  if (!g) return PPS_EINVAL;
  if (!live_node(g, plane, NODE_PLANE)) return fail(g, PPS_EINVAL, "plane prior: unknown plane id");
  return add_factor(g, F_PLANE_PRIOR, plane, -1, meas4, 4, ut, 6, fid);
}

int pps_set_measurement(pps_graph* g, int fid, const double meas4[4]) { return pps_set_measurements(g, 1, &fid, meas4); }

int pps_set_measurements(pps_graph* g, int n, const int* fids, const double* meas4) {
  if (!g || !fids || !meas4 || n < 0) return PPS_EINVAL;
  if (g->dev_meas_newer) { int rc = download_measurements(g); if (rc != PPS_OK) return rc; }
  for (int i = 0; i < n; i++) {
    const int fid = fids[i];
    if (fid < 0 || fid >= (int)g->factors.size() || g->factors[fid].deleted) return fail(g, PPS_EINVAL, "set_measurement: unknown factor id");
    HostFactor& f = g->factors[fid];
    if (f.type != F_PLANE_OBS && f.type != F_PLANE_PRIOR) return fail(g, PPS_EINVAL, "set_measurement: not a plane factor");
    for (int k = 0; k < 4; k++) {
      if (!std::isfinite(meas4[4 * i + k])) return fail(g, PPS_EINVAL, "non-finite measurement");
      f.meas[k] = meas4[4 * i + k];
    }
    normalize4(f.meas);
  }
  g->meas_dirty = true; g->pk_meas_ok = false;
  return PPS_OK;
}

int pps_remove_factor(pps_graph* g, int fid) {
  if (!g) return PPS_EINVAL;
  if (fid < 0 || fid >= (int)g->factors.size() || g->factors[fid].deleted) return fail(g, PPS_EINVAL, "remove_factor: unknown id");
  g->factors[fid].deleted = true;
  g->n_removals++;
  g->grown_only = false; g->grown_only_upload = false;
  g->n_live_factors--;
  g->n_live_type[g->factors[fid].type]--;
  if (g->factors[fid].type == F_PLANE_OBS && g->factors[fid].repop) g->n_live_repop--;
  g->dim_measure -= kFDim[g->factors[fid].type];
  g->topo_dirty = true; g->analysis_stale = true;
  return PPS_OK;
}

int pps_remove_node(pps_graph* g, int nid) {
  if (!g) return PPS_EINVAL;
  if (nid < 0 || nid >= (int)g->nodes.size() || g->nodes[nid].deleted) return fail(g, PPS_EINVAL, "remove_node: unknown id");
  if (g->dev_values_newer) { int rc = download_state(g); if (rc != PPS_OK) return rc; }
  for (size_t i = 0; i < g->factors.size(); i++) {
    HostFactor& f = g->factors[i];
    if (!f.deleted && (f.a == nid || f.b == nid)) pps_remove_factor(g, (int)i);
  }
  g->nodes[nid].deleted = true;
  g->n_removals++;
  g->grown_only = false; g->grown_only_upload = false;
  g->n_live_nodes--;
  g->dim_nodes -= g->nodes[nid].type == NODE_POSE ? 6 : 3;
  g->topo_dirty = true; g->analysis_stale = true; g->host_values_newer = true;
  return PPS_OK;
}

int pps_num_nodes(const pps_graph* g, int* n) { if (!g || !n) return PPS_EINVAL; *n = g->n_live_nodes; return PPS_OK; }
int pps_num_factors(const pps_graph* g, int* n) { if (!g || !n) return PPS_EINVAL; *n = g->n_live_factors; return PPS_OK; }

int pps_get_pose(pps_graph* g, int id, double tq[7]) {
  if (!g || !tq) return PPS_EINVAL;
  if (!live_node(g, id, NODE_POSE)) return fail(g, PPS_EINVAL, "get_pose: unknown id");
  int rc = download_state(g); if (rc != PPS_OK) return rc;
  memcpy(tq, g->nodes[id].v, 7 * sizeof(double));
  return PPS_OK;
}
int pps_get_plane(pps_graph* g, int id, double abcd[4]) {
  if (!g || !abcd) return PPS_EINVAL;
  if (!live_node(g, id, NODE_PLANE)) return fail(g, PPS_EINVAL, "get_plane: unknown id");
  int rc = download_state(g); if (rc != PPS_OK) return rc;
  memcpy(abcd, g->nodes[id].v, 4 * sizeof(double));
  return PPS_OK;
}
int pps_set_pose(pps_graph* g, int id, const double tq[7]) {
  if (!g || !tq) return PPS_EINVAL;
  if (!live_node(g, id, NODE_POSE)) return fail(g, PPS_EINVAL, "set_pose: unknown id");
  int rc = download_state(g); if (rc != PPS_OK) return rc;
  memcpy(g->nodes[id].v, tq, 7 * sizeof(double));
  g->host_values_newer = true;
  return PPS_OK;
}
int pps_set_plane(pps_graph* g, int id, const double abcd[4]) {
  if (!g || !abcd) return PPS_EINVAL;
  if (!live_node(g, id, NODE_PLANE)) return fail(g, PPS_EINVAL, "set_plane: unknown id");
  int rc = download_state(g); if (rc != PPS_OK) return rc;
  memcpy(g->nodes[id].v, abcd, 4 * sizeof(double));
  normalize4(g->nodes[id].v);
  g->host_values_newer = true;
  return PPS_OK;
}

static int get_bulk(pps_graph* g, int type, int n, const int* ids, double* out, int w) {
  if (!g || !out || n < 0) return PPS_EINVAL;
  int rc = download_state(g); if (rc != PPS_OK) return rc;
  if (ids) {
    for (int i = 0; i < n; i++) {
      if (!live_node(g, ids[i], type)) return fail(g, PPS_EINVAL, "bulk get: unknown id");
      memcpy(out + (size_t)w * i, g->nodes[ids[i]].v, w * sizeof(double));
    }
  } else {
    int k = 0;
    for (size_t i = 0; i < g->nodes.size() && k < n; i++)
      if (!g->nodes[i].deleted && g->nodes[i].type == type) { memcpy(out + (size_t)w * k, g->nodes[i].v, w * sizeof(double)); k++; }
    if (k != n) return fail(g, PPS_EINVAL, "bulk get: count mismatch");
  }
  return PPS_OK;
}
int pps_get_poses(pps_graph* g, int n, const int* ids, double* out) { return get_bulk(g, NODE_POSE, n, ids, out, 7); }
int pps_get_planes(pps_graph* g, int n, const int* ids, double* out) { return get_bulk(g, NODE_PLANE, n, ids, out, 4); }

int pps_save_state(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  const DevGraph& d = g->dev;
  if (!g->snap_pose || g->snap_version != g->upload_version) {
    // (re)allocate with the current leading dimensions; owned by the allocation list of this upload
    rc = dev_alloc(g, &g->snap_pose, (size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld); if (rc != PPS_OK) return rc;
    g->snap_plane = g->snap_pose + (size_t)7 * d.pose_ld;
    g->snap_version = g->upload_version;
  }
  HIP_TRY(g, hipMemcpyAsync(g->snap_pose, d.pose_est, ((size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld) * 8, hipMemcpyDeviceToDevice, g->stream));
  return PPS_OK;
}

int pps_restore_state(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  if (!g->snap_pose || g->snap_version != g->upload_version || g->topo_dirty) return fail(g, PPS_ESTATE, "no snapshot for the current topology");
  if (g->host_values_newer) return fail(g, PPS_ESTATE, "host values were modified after the snapshot");
  const DevGraph& d = g->dev;
  HIP_TRY(g, hipMemcpyAsync(d.pose_est, g->snap_pose, ((size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld) * 8, hipMemcpyDeviceToDevice, g->stream));
  g->dev_values_newer = true; g->pin_holds_est = false;
  return PPS_OK;
}

int pps_get_stats(const pps_graph* g, pps_stats* out) {
  if (!g || !out) return PPS_EINVAL;
  *out = g->stats;
  out->n_poses = out->n_planes = 0;
  for (const auto& n : g->nodes) if (!n.deleted) (n.type == NODE_POSE ? out->n_poses : out->n_planes)++;
  out->n_factors = g->n_live_factors;
  out->dim_nodes = g->dim_nodes; out->dim_measure = g->dim_measure;
  return PPS_OK;
}

int pps_get_trace(const pps_graph* g, int cap, double* lambda, double* chi2, int* accepted, int* n) {
  if (!g || !n) return PPS_EINVAL;
  *n = (int)g->tr_lambda.size();
  for (int i = 0; i < *n && i < cap; i++) {
    if (lambda) lambda[i] = g->tr_lambda[i];
    if (chi2) chi2[i] = g->tr_chi2[i];
    if (accepted) accepted[i] = g->tr_acc[i];
  }
  return PPS_OK;
}

int pps_set_profiling(pps_graph* g, int level) { if (!g) return PPS_EINVAL; g->profiling = level < 0 ? 0 : level; return PPS_OK; }

int pps_factor_shape(const pps_graph* g, int fid, int* dim, int* cols) {
  if (!g) return PPS_EINVAL;
  if (fid < 0 || fid >= (int)g->factors.size() || g->factors[fid].deleted) return PPS_EINVAL;
  const HostFactor& f = g->factors[fid];
  if (dim) *dim = kFDim[f.type];
  if (cols) *cols = (g->nodes[f.a].type == NODE_POSE ? 6 : 3) + (f.b >= 0 ? (g->nodes[f.b].type == NODE_POSE ? 6 : 3) : 0);
  return PPS_OK;
}

int pps_eval_factor(pps_graph* g, int fid, int mode, double* J, double* r) {
  if (!g || !J || !r) return PPS_EINVAL;
  if (fid < 0 || fid >= (int)g->factors.size() || g->factors[fid].deleted) return fail(g, PPS_EINVAL, "eval_factor: unknown id");
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  rc = linpoint_from_estimate(g); if (rc != PPS_OK) return rc;
  HIP_TRY(g, launch_linearize(g->dev, mode, true, g->stream));
  const HostFactor& f = g->factors[fid];
  const int m = kFDim[f.type];
  const int da = g->nodes[f.a].type == NODE_POSE ? 6 : 3;
  const int db = f.b >= 0 ? (g->nodes[f.b].type == NODE_POSE ? 6 : 3) : 0;
  const int64_t base[4] = {g->dev.joff_pp, g->dev.joff_odo, g->dev.joff_obs, g->dev.joff_lp};
  std::vector<double> buf(kJSize[f.type]);
  HIP_TRY(g, hipMemcpyAsync(buf.data(), g->dev.J + base[f.type] + (int64_t)f.slot * kJSize[f.type], buf.size() * 8,
                            hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  const int cols = da + db;
  for (int i = 0; i < m; i++) {
    for (int j = 0; j < da; j++) J[i * cols + j] = buf[i * da + j];
    for (int j = 0; j < db; j++) J[i * cols + da + j] = buf[m * da + i * db + j];
    r[i] = buf[m * (da + db) + i];
  }
  return PPS_OK;
}

int pps_analyze(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  if (g->n_live_nodes == 0) return fail(g, PPS_ESTATE, "empty graph");
  // run_analysis re-assigns the node / factor slots; anything that is newer on the device still lives in the OLD slot layout
  // and has to come home first (a refresh or a solve followed by an edit, then this call)
  if (g->dev_ready) {
    int rc = download_state(g); if (rc != PPS_OK) return rc;
    rc = download_measurements(g); if (rc != PPS_OK) return rc;
    g->topo_dirty = true;            // the device arrays no longer match the slot tables: the next solve uploads again
  }
  return run_analysis(g);
}

int pps_analysis_reuse(const pps_graph* g, int* fronts_kept, int* fronts_total) {
  if (!g) return PPS_EINVAL;
  analysis_cache_stats(g->acache, fronts_kept, fronts_total);
  return PPS_OK;
}

int pps_analysis_kept(const pps_graph* g, int kept[6]) {
  if (!g || !kept) return PPS_EINVAL;
  const Analysis::Kept& k = g->an.kept;
  kept[0] = k.fronts; kept[1] = k.fronts_lists; kept[2] = k.blocks; kept[3] = k.segs; kept[4] = k.contribs; kept[5] = k.nd_segs;
  return PPS_OK;
}

int pps_analysis_dump(pps_graph* g, int64_t cap, int32_t* out, int64_t* needed) {
  if (!g || !needed) return PPS_EINVAL;
  if (!g->analyzed || g->analysis_stale) { int rc = pps_analyze(g); if (rc != PPS_OK) return rc; }
  std::vector<int32_t> v;
  dump_analysis(g->an, v);
  // append the compact-id tables the tests need: node id -> compact id, factor id -> joff
  v.push_back((int32_t)g->nodes.size());
  for (const auto& n : g->nodes) v.push_back(n.deleted ? -1 : n.compact);
  int64_t base[4]; j_bases(g, base, nullptr);
  v.push_back((int32_t)g->factors.size());
  for (const auto& f : g->factors) v.push_back(f.deleted ? -1 : (int32_t)(base[f.type] + (int64_t)f.slot * kJSize[f.type]));
  int64_t pbase[4]; p_bases(g, pbase, nullptr);                  // ... and factor id -> offset of its product record
  v.push_back((int32_t)g->factors.size());
  for (const auto& f : g->factors) v.push_back(f.deleted ? -1 : (int32_t)(pbase[f.type] + (int64_t)f.slot * kPSize[f.type]));
  v.push_back((int32_t)g->an.P_size);
  *needed = (int64_t)v.size();
  if (out && cap >= (int64_t)v.size()) memcpy(out, v.data(), v.size() * sizeof(int32_t));
  return PPS_OK;
}

// K1 alone on the solver's stream: `iters` back-to-back sweeps of the handle's graph between two HIP events
// (an event pair around ONE ~10 us launch also measures the command processor's event handling, about as long again)
int pps_debug_front_factor(int tiles, int strip, int p, int b, const double* A, double* L, double* U, double* not_pd) {
  if (!A || !L || !U) return PPS_EINVAL;
  const int rc = pps::debug_front_factor(tiles, strip, p, b, A, L, U, not_pd);
  return rc == 0 ? PPS_OK : (rc < 0 ? PPS_EINVAL : PPS_EHIP);
}

int pps_debug_exmap(int kind, int n, const double* x, const double* delta, double* out) {
  if (!x || !delta || !out) return PPS_EINVAL;
  const int rc = pps::debug_exmap(kind, n, x, delta, out);
  return rc == 0 ? PPS_OK : (rc < 0 ? PPS_EINVAL : PPS_EHIP);
}

int pps_time_linearize(pps_graph* g, int mode, int iters, double* sec_per_launch) {
  if (!g || iters < 1 || !sec_per_launch) return PPS_EINVAL;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  rc = linpoint_from_estimate(g); if (rc != PPS_OK) return rc;
  for (int k = 0; k < 3; k++) HIP_TRY(g, launch_linearize(g->dev, mode, false, g->stream));
  HIP_TRY(g, hipEventRecord(g->ev[0], g->stream));
  for (int k = 0; k < iters; k++) HIP_TRY(g, launch_linearize(g->dev, mode, false, g->stream));
  HIP_TRY(g, hipEventRecord(g->ev[1], g->stream));
  HIP_TRY(g, hipEventSynchronize(g->ev[1]));
  float ms = 0;
  HIP_TRY(g, hipEventElapsedTime(&ms, g->ev[0], g->ev[1]));
  *sec_per_launch = 1e-3 * ms / iters;
  return PPS_OK;
}

int pps_bench_sweep(pps_graph* g, int mode, int replicas, int iters, double* sec_per_sweep, int64_t* n_plane_edges,
                    int64_t* n_odo_edges) {
  if (!g || replicas < 1 || iters < 1 || !sec_per_sweep) return PPS_EINVAL;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  rc = linpoint_from_estimate(g); if (rc != PPS_OK) return rc;
  if (g->dev.n_obs_fixed != g->dev.n_obs) return fail(g, PPS_ESTATE, "bench_sweep: graph holds re-popping plane edges (Factor2)");
  DevGraph d = g->dev;   // shallow copy with replicated edge arrays
  std::vector<void*> tmp;
  auto rep = [&](auto** ptr, size_t count_per) -> int {
    using T = std::remove_pointer_t<std::remove_pointer_t<decltype(ptr)>>;
    if (count_per == 0) return PPS_OK;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, count_per * sizeof(T) * (size_t)replicas);
    if (e != hipSuccess) return hip_fail(g, e, "hipMalloc(bench)");
    tmp.push_back(p);
    for (int r = 0; r < replicas; r++) {
      e = hipMemcpyAsync(static_cast<char*>(p) + (size_t)r * count_per * sizeof(T), *ptr, count_per * sizeof(T),
                         hipMemcpyDeviceToDevice, g->stream);
      if (e != hipSuccess) return hip_fail(g, e, "hipMemcpyAsync(bench)");
    }
    *ptr = static_cast<T*>(p);
    return PPS_OK;
  };
  auto cleanup = [&]() { (void)hipStreamSynchronize(g->stream); for (void* p : tmp) (void)hipFree(p); };
#define BT(x) do { rc = (x); if (rc != PPS_OK) { cleanup(); return rc; } } while (0)
  BT(rep(&d.obs_pose, (size_t)d.n_obs)); BT(rep(&d.obs_plane, (size_t)d.n_obs));
  BT(rep(&d.obs_meas, (size_t)4 * d.obs_ld)); BT(rep(&d.obs_w, (size_t)6 * d.obs_ld));
  BT(rep(&d.odo_a, (size_t)d.n_odo)); BT(rep(&d.odo_b, (size_t)d.n_odo));
  BT(rep(&d.odo_meas, (size_t)6 * d.odo_ld)); BT(rep(&d.odo_w, (size_t)21 * d.odo_ld));
#undef BT
  const size_t slab = (size_t)d.n_obs * 30 + (size_t)d.n_odo * 78;
  double* Jbig = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&Jbig), slab * sizeof(double) * (size_t)replicas);
  if (e != hipSuccess) { cleanup(); return hip_fail(g, e, "hipMalloc(Jbig)"); }
  tmp.push_back(Jbig);
  e = launch_sweep_bench(d, mode, replicas, Jbig, -1, g->stream);   // warm-up
  if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  float total_ms = 0, part_ms[2] = {0, 0};
  for (int it = 0; it < iters && e == hipSuccess; it++) {
    (void)hipEventRecord(g->ev[0], g->stream);
    e = launch_sweep_bench(d, mode, replicas, Jbig, -1, g->stream);
    (void)hipEventRecord(g->ev[1], g->stream);
    if (e == hipSuccess) e = hipEventSynchronize(g->ev[1]);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, g->ev[0], g->ev[1]);
    total_ms += ms;
    for (int part = 0; part < 2 && e == hipSuccess; part++) {       // each launch on its own
      (void)hipEventRecord(g->ev[0], g->stream);
      e = launch_sweep_bench(d, mode, replicas, Jbig, part, g->stream);
      (void)hipEventRecord(g->ev[1], g->stream);
      if (e == hipSuccess) e = hipEventSynchronize(g->ev[1]);
      (void)hipEventElapsedTime(&ms, g->ev[0], g->ev[1]);
      part_ms[part] += ms;
    }
  }
  cleanup();
  if (e != hipSuccess) return hip_fail(g, e, "sweep bench");
  sec_per_sweep[1] = 1e-3 * part_ms[0] / iters;
  sec_per_sweep[2] = 1e-3 * part_ms[1] / iters;
  *sec_per_sweep = 1e-3 * total_ms / iters;
  if (n_plane_edges) *n_plane_edges = (int64_t)d.n_obs * replicas;
  if (n_odo_edges) *n_odo_edges = (int64_t)d.n_odo * replicas;
  return PPS_OK;
}

}  // extern "C"
