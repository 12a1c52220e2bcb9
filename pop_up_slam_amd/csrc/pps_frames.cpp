// pps_frames.cpp -- what the mapper does around the solve: registered frames whose plane measurements are re-derived on the
// device (Mapper_mono::update_plane_measurement), landmark records and data association (findClosestPlane), re-projection of
// stored polygon vertices (reproj_to_newplane).
#include "pps_graph.h"

using namespace pps;
using namespace pps_impl;

extern "C" {

int pps_frames_set_calibration(pps_graph* g, const float invK[9]) {
  if (!g || !invK) return PPS_EINVAL;
  memcpy(g->frames_invK, invK, sizeof g->frames_invK);
  return PPS_OK;
}

int pps_frames_add(pps_graph* g, int pose_id, int n_seg, const float* seg2d, const int* fids, int* frame_id) {
  if (!g || n_seg < 0 || !fids || (n_seg > 0 && !seg2d)) return PPS_EINVAL;
  if (!live_node(g, pose_id, NODE_POSE)) return fail(g, PPS_EINVAL, "frames_add: unknown pose id");
  for (int j = 0; j <= n_seg; j++) {
    const int fid = fids[j];
    if (fid < 0) continue;
    if (fid >= (int)g->factors.size() || g->factors[fid].deleted || g->factors[fid].type != F_PLANE_OBS || g->factors[fid].a != pose_id)
      return fail(g, PPS_EINVAL, "frames_add: fid is not a plane observation of this pose");
  }
  const int f = (int)g->fr_pose.size();
  g->fr_pose.push_back(pose_id);
  for (int k = 0; k < 4 * n_seg; k++) g->fr_seg.push_back(seg2d[k]);
  g->fr_seg_off.push_back(g->fr_seg_off.back() + n_seg);
  for (int j = 0; j <= n_seg; j++) { g->fr_item_frame.push_back(f); g->fr_item_plane.push_back(j); g->fr_item_fid.push_back(fids[j]); }
  g->frames_dirty = true;
  if (frame_id) *frame_id = f;
  return PPS_OK;
}

int pps_refresh_measurements(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  if (g->fr_item_frame.empty()) return PPS_OK;
  if (g->frames_dirty) {
    // factor and pose slots only append while nothing is removed: the tables are extended by the new frame's items
    std::vector<int>& slot = g->fr_slot; std::vector<int>& pslot = g->fr_pslot;
    const bool inc = g->fr_cache_removals == g->n_removals && slot.size() <= g->fr_item_fid.size() && pslot.size() <= g->fr_pose.size();
    if (!inc) { slot.clear(); pslot.clear(); }
    const size_t i0 = slot.size(), f0 = pslot.size();
    slot.resize(g->fr_item_fid.size()); pslot.resize(g->fr_pose.size());
    for (size_t i = i0; i < slot.size(); i++) {
      const int fid = g->fr_item_fid[i];
      slot[i] = (fid >= 0 && !g->factors[fid].deleted && !g->factors[fid].repop) ? g->factors[fid].slot : -1;
      if (slot[i] >= 0 && g->nodes[g->fr_pose[g->fr_item_frame[i]]].deleted) slot[i] = -1;
    }
    for (size_t f = f0; f < pslot.size(); f++) pslot[f] = g->nodes[g->fr_pose[f]].deleted ? 0 : g->nodes[g->fr_pose[f]].slot;
    g->fr_cache_removals = g->n_removals;
    // tables live in the allocation list of the current upload; older copies are simply abandoned until then.  (The topology
    // upload may still be copying out of the pinned mirror and the patch buffer this is about to write.)
    if (g->up_inflight) { HIP_TRY(g, hipStreamSynchronize(g->stream)); g->up_inflight = false; }
    // The first table upload after an upload_all goes to the slots the previous layout's first table upload went to (same number of
    // uploads in front of it: same cursor), which nothing has written since: the entries it sent are in the mirror.
    g->verify_hints = g->sw.verify_upload;
    const size_t c0 = g->up_cursor;
    const bool first = g->fr_hint_version != g->upload_version;
    const bool hint = inc && first && !g->up_unknown && g->fr_hint_version + 1 == g->upload_version && g->fr_hint_cursor == c0 &&
                      g->fr_hint_items <= slot.size() && g->fr_hint_frames <= pslot.size() && i0 >= g->fr_hint_items && f0 >= g->fr_hint_frames;
    const size_t hi = hint ? g->fr_hint_items : 0, hf = hint ? g->fr_hint_frames : 0;
    rc = dev_upload(g, &g->d_item_frame, g->fr_item_frame, hi); if (rc != PPS_OK) return rc;
    rc = dev_upload(g, &g->d_item_plane, g->fr_item_plane, hi); if (rc != PPS_OK) return rc;
    rc = dev_upload(g, &g->d_item_slot, slot, hi); if (rc != PPS_OK) return rc;
    rc = dev_upload(g, &g->d_frame_pose_slot, pslot, hf); if (rc != PPS_OK) return rc;
    rc = dev_upload(g, &g->d_frame_seg_off, g->fr_seg_off, hf ? hf + 1 : 0); if (rc != PPS_OK) return rc;
    rc = dev_upload(g, &g->d_fr_seg, g->fr_seg, hf ? 4 * (size_t)g->fr_seg_off[hf] : 0); if (rc != PPS_OK) return rc;
    if (first) { g->fr_hint_version = g->upload_version; g->fr_hint_cursor = c0; g->fr_hint_items = slot.size(); g->fr_hint_frames = pslot.size(); }
    rc = flush_uploads(g); if (rc != PPS_OK) return rc;
    rc = verify_uploads(g, "frames"); if (rc != PPS_OK) return rc;
    g->up_inflight = true;                                  // (whoever writes the mirror next waits for this copy)
    g->frames_dirty = false;
  }
  RefreshArgs a{};
  a.n_items = (int)g->fr_item_frame.size();
  a.item_frame = g->d_item_frame; a.item_plane = g->d_item_plane; a.item_slot = g->d_item_slot;
  a.frame_pose_slot = g->d_frame_pose_slot; a.frame_seg_off = g->d_frame_seg_off; a.seg2d = g->d_fr_seg;
  memcpy(a.invK, g->frames_invK, sizeof a.invK);
  a.pose_est = g->dev.pose_est; a.pose_ld = g->dev.pose_ld;
  a.obs_meas = g->dev.obs_meas; a.n_obs = g->dev.n_obs; a.obs_ld = g->dev.obs_ld;
  HIP_TRY(g, launch_refresh_measurements(a, g->stream));
  g->dev_meas_newer = true;
  return PPS_OK;
}

int pps_get_measurement(pps_graph* g, int fid, double meas4[4]) {
  if (!g || !meas4) return PPS_EINVAL;
  if (fid < 0 || fid >= (int)g->factors.size() || g->factors[fid].deleted) return fail(g, PPS_EINVAL, "get_measurement: unknown factor id");
  const HostFactor& f = g->factors[fid];
  if (f.type != F_PLANE_OBS && f.type != F_PLANE_PRIOR) return fail(g, PPS_EINVAL, "get_measurement: not a plane factor");
  if (g->dev_meas_newer) { int rc = download_measurements(g); if (rc != PPS_OK) return rc; }
  memcpy(meas4, f.meas, 4 * sizeof(double));
  return PPS_OK;
}


// ---- plane data association (Mapper_mono::findClosestPlane, src/Mapping.cpp:256-397) ----

void pps_assoc_default_params(pps_assoc_params* p) {
  if (!p) return;
  p->edge_asso_2ddist = 50; p->edge_asso_planedist = 4; p->edge_asso_proj = 0.5; p->edge_asso_angle = 60.0;   // Mapping.h:72-76
  p->assoc_near_frames = 5;
}

int pps_landmark_update(pps_graph* g, int plane_id, int frame_plane_indice, int frame_seq_id, const float seg2d[4],
                        const float seg3d_xy[4]) {
  if (!g) return PPS_EINVAL;
  if (!live_node(g, plane_id, NODE_PLANE)) return fail(g, PPS_EINVAL, "landmark_update: unknown plane id");
  auto it = g->lm_of_plane.find(plane_id);
  int idx;
  if (it == g->lm_of_plane.end()) {
    idx = (int)g->lms.size();
    g->lms.push_back(pps_graph::Landmark{plane_id, 0, 0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}});
    g->lm_of_plane[plane_id] = idx;
  } else idx = it->second;
  pps_graph::Landmark& L = g->lms[idx];
  L.fpi = frame_plane_indice; L.seq = frame_seq_id;
  for (int k = 0; k < 4; k++) { L.seg2d[k] = seg2d ? seg2d[k] : 0.f; L.seg3d[k] = seg3d_xy ? seg3d_xy[k] : 0.f; }
  g->lms_dirty = true;
  return PPS_OK;
}

int pps_landmark_set_merged(pps_graph* g, int plane_id) {
  if (!g) return PPS_EINVAL;
  auto it = g->lm_of_plane.find(plane_id);
  if (it == g->lm_of_plane.end()) return fail(g, PPS_EINVAL, "landmark_set_merged: plane id is not a landmark");
  g->lms[it->second].deleted = 1;
  g->lms_dirty = true;
  return PPS_OK;
}

int pps_find_closest_planes(pps_graph* g, const double est_pose[7], int frame_seq_id, int n, const double* planes_local,
                            const int* frame_plane_indice, const float* seg2d, const float* seg3d_xy,
                            const pps_assoc_params* prm, int* best_plane_id, double* best_err) {
  if (!g || !est_pose || n < 0 || (n > 0 && (!planes_local || !frame_plane_indice || !seg2d || !seg3d_xy)) || !best_plane_id || !best_err)
    return PPS_EINVAL;
  pps_assoc_params P;
  if (prm) P = *prm; else pps_assoc_default_params(&P);
  if (n == 0) return PPS_OK;
  const int nl = (int)g->lms.size();
  if (nl == 0) { for (int i = 0; i < n; i++) { best_plane_id[i] = -1; best_err[i] = -1.0; } return PPS_OK; }
  int rc = ensure_device(g);
  if (rc != PPS_OK) return rc;
  HIP_TRY(g, hipSetDevice(g->props.device));
  // landmark planes: straight from the solver state when it is current, else a packed copy of the host values
  const bool state_current = g->dev_ready && !g->topo_dirty && !g->host_values_newer && g->dev.n_plane > 0;
  AssocArgs a{};
  if (!state_current) {
    if (g->dev_values_newer) { rc = download_state(g); if (rc != PPS_OK) return rc; }
    if ((size_t)nl > g->d_lm_planes_cap) {
      if (g->d_lm_planes) (void)hipFree(g->d_lm_planes);
      g->d_lm_planes_cap = std::max<size_t>(256, 2 * (size_t)nl);
      HIP_TRY(g, hipMalloc(reinterpret_cast<void**>(&g->d_lm_planes), 4 * g->d_lm_planes_cap * sizeof(double)));
    }
    std::vector<double> pl(4 * (size_t)nl, 0.0);
    for (int i = 0; i < nl; i++) {
      const HostNode& nd = g->nodes[g->lms[i].plane_id];
      for (int k = 0; k < 4; k++) pl[(size_t)k * nl + i] = nd.v[k];
    }
    HIP_TRY(g, hipMemcpyAsync(g->d_lm_planes, pl.data(), pl.size() * sizeof(double), hipMemcpyHostToDevice, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));   // `pl` leaves scope
    a.plane_est = g->d_lm_planes; a.plane_ld = nl;
  } else {
    a.plane_est = g->dev.plane_est; a.plane_ld = g->dev.plane_ld;
  }
  if (g->lms_dirty || g->lms_upload_version != (state_current ? g->upload_version : -2)) {
    if ((size_t)nl > g->d_lms_cap) {
      HIP_TRY(g, hipStreamSynchronize(g->stream));
      if (g->d_lms) (void)hipFree(g->d_lms);
      if (g->h_lms) (void)hipHostFree(g->h_lms);
      g->d_lms = nullptr; g->h_lms = nullptr;
      g->d_lms_cap = std::max<size_t>(256, 2 * (size_t)nl);
      HIP_TRY(g, hipMalloc(reinterpret_cast<void**>(&g->d_lms), g->d_lms_cap * sizeof(AssocLandmark)));
      HIP_TRY(g, hipHostMalloc(reinterpret_cast<void**>(&g->h_lms), g->d_lms_cap * sizeof(AssocLandmark), hipHostMallocDefault));
    }
    AssocLandmark* rec = g->h_lms;       // (pinned; the copy below has ended when this call returns: the call ends with a stream synchronisation)
    for (int i = 0; i < nl; i++) {
      const pps_graph::Landmark& L = g->lms[i];
      const HostNode& nd = g->nodes[L.plane_id];
      rec[i].plane_slot = nd.deleted ? -1 : (state_current ? nd.slot : i);
      rec[i].frame_plane_indice = L.fpi; rec[i].frame_seq_id = L.seq; rec[i].deleted = L.deleted;
      memcpy(rec[i].seg2d, L.seg2d, sizeof L.seg2d); memcpy(rec[i].seg3d, L.seg3d, sizeof L.seg3d);
    }
    HIP_TRY(g, hipMemcpyAsync(g->d_lms, rec, (size_t)nl * sizeof(AssocLandmark), hipMemcpyHostToDevice, g->stream));
    g->lms_dirty = false;
    g->lms_upload_version = state_current ? g->upload_version : -2;
  }
  if ((size_t)n > g->d_q_cap) {
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    if (g->d_queries) (void)hipHostFree(g->d_queries);
    if (g->d_results) (void)hipHostFree(g->d_results);
    g->d_queries = nullptr; g->d_results = nullptr;
    g->d_q_cap = std::max<size_t>(64, 2 * (size_t)n);
    HIP_TRY(g, hipHostMalloc(reinterpret_cast<void**>(&g->d_queries), g->d_q_cap * sizeof(AssocQuery), hipHostMallocDefault));
    HIP_TRY(g, hipHostMalloc(reinterpret_cast<void**>(&g->d_results), g->d_q_cap * sizeof(AssocResult), hipHostMallocDefault));
  }
  // the queries of a frame (a few hundred bytes) and their results live in pinned host memory: the kernel reads and writes them over the bus,
  // the call synchronises once (round 6; it was a pageable copy each way)
  AssocQuery* q = g->d_queries;
  for (int i = 0; i < n; i++) {
    memcpy(q[i].plane_local, planes_local + 4 * i, 4 * sizeof(double));
    memcpy(q[i].seg2d, seg2d + 4 * i, 4 * sizeof(float)); memcpy(q[i].seg3d, seg3d_xy + 4 * i, 4 * sizeof(float));
    q[i].frame_plane_indice = frame_plane_indice[i]; q[i].frame_seq_id = frame_seq_id;
    memset(q[i].pad, 0, sizeof q[i].pad);
  }
  a.n_queries = n; a.n_landmarks = nl; a.queries = g->d_queries; a.landmarks = g->d_lms; a.results = g->d_results;
  memcpy(a.pose, est_pose, sizeof a.pose);
  a.edge_asso_2ddist = P.edge_asso_2ddist; a.edge_asso_planedist = P.edge_asso_planedist;
  a.edge_asso_proj = P.edge_asso_proj; a.edge_asso_angle = P.edge_asso_angle; a.assoc_near_frames = P.assoc_near_frames;
  HIP_TRY(g, launch_assoc(a, g->stream));
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  const AssocResult* r = g->d_results;
  for (int i = 0; i < n; i++) {
    best_plane_id[i] = r[i].best >= 0 ? g->lms[r[i].best].plane_id : -1;
    best_err[i] = r[i].err;
  }
  return PPS_OK;
}


// ---- graph text format: Slam::save (Slam.cpp:84-89) -> Graph::write (Graph.h:120-131) ----
//   factor line   <name> <node ids> <measure> {sqrtinf upper triangle, row-first}   (Factor.h:148-155,208-211,169-187)
//   node line     <Type>_Node <id> <value>                                          (Node.h:148-153)
//   Pose3d (x, y, z; yaw, pitch, roll) (Pose3d.h:169-172);  Plane3d (a, b, c; d) (isam_plane3d.h:190-192)
// The plane prior prints as "Pose3d_Factor" like the pose prior (constructor name, isam_plane3d.h:438).
// precision <= 0 selects the ostream default of the reference (6 significant digits, lossy); 17 round-trips.

// Mapper_mono::reproj_to_newplane (src/Mapping.cpp:609-632): polygon vertices onto the optimised planes
int pps_reproject_points(pps_graph* g, int n, const int* plane_ids, const float* pts_xyz, float* out_xyz) {
  if (!g || n < 0 || (n > 0 && (!plane_ids || !pts_xyz || !out_xyz))) return PPS_EINVAL;
  if (n == 0) return PPS_OK;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  // one pinned block [slots | points in | points out] that the kernel reads and writes over the bus (every element once): no allocation, no
  // copy and one synchronisation per call (up to round 5: three hipMalloc / hipFree pairs and three copies per call)
  const size_t off_in = ((size_t)n * sizeof(int) + 15) & ~size_t(15), off_out = off_in + (((size_t)3 * n * sizeof(float) + 15) & ~size_t(15));
  const size_t need = off_out + (size_t)3 * n * sizeof(float);
  if (need > g->rp_cap) {
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    if (g->rp_pin) (void)hipHostFree(g->rp_pin);
    g->rp_pin = nullptr; g->rp_cap = 0;
    const size_t cap = std::max<size_t>(1 << 14, 2 * need);
    HIP_TRY(g, hipHostMalloc(reinterpret_cast<void**>(&g->rp_pin), cap, hipHostMallocDefault));
    g->rp_cap = cap;
  }
  int* slot = reinterpret_cast<int*>(g->rp_pin);
  float* p_in = reinterpret_cast<float*>(g->rp_pin + off_in);
  float* p_out = reinterpret_cast<float*>(g->rp_pin + off_out);
  for (int i = 0; i < n; i++) slot[i] = live_node(g, plane_ids[i], NODE_PLANE) ? g->nodes[plane_ids[i]].slot : -1;
  std::memcpy(p_in, pts_xyz, (size_t)3 * n * sizeof(float));
  hipError_t e = launch_reproject(n, slot, p_in, g->dev.plane_est, g->dev.plane_ld, p_out, g->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  if (e == hipSuccess) std::memcpy(out_xyz, p_out, (size_t)3 * n * sizeof(float));
  if (e != hipSuccess) return hip_fail(g, e, "pps_reproject_points");
  return PPS_OK;
}


}  // extern "C"
