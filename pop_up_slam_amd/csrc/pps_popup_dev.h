// pps_popup_dev.h -- argument blocks shared by the pop-up kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>

namespace pps {

struct PopupParams {
  float invK[9];
  float T[16];          // T_wc (camera -> world), row-major
  int width, height, step;
  float depth_thre, ceiling_thre;
};

// Mapper_mono::update_plane_measurement on the device: one item per (frame, plane)
struct RefreshArgs {
  int n_items;
  const int* item_frame;        // frame index
  const int* item_plane;        // 0 = ground, j>=1 = segment j-1 of the frame
  const int* item_slot;         // slot in the plane-observation arrays (-1 = skip)
  const int* frame_pose_slot;   // frame -> pose slot
  const int* frame_seg_off;     // frame -> first segment
  const float* seg2d;           // all segments, 4 floats each
  float invK[9];
  const double* pose_est; int pose_ld;
  double* obs_meas; int n_obs, obs_ld;
};

hipError_t launch_refresh_measurements(const RefreshArgs& a, hipStream_t st);

}  // namespace pps

namespace pps {

// Mapper_mono::findClosestPlane on the device (src/Mapping.cpp:256-397): one workgroup per query plane,
// the landmarks are scored in parallel and reduced with the sequential loop's first-wins semantics.
struct AssocLandmark {       // 48 bytes, one record per all_landmarks entry
  int plane_slot;            // slot of the landmark's plane node in plane_est (-1: node gone)
  int frame_plane_indice;    // 0 = ground, >= 1 wall
  int frame_seq_id;
  int deleted;               // deteted_by_merge
  float seg2d[4];            // plane_bound_close_2D_polys columns 0,1
  float seg3d[4];            // plane_bound_close_3D_polys columns 0,1 (x,y)
};
struct AssocQuery {          // 88 bytes
  double plane_local[4];
  float seg2d[4];
  float seg3d[4];
  int frame_plane_indice;
  int frame_seq_id;
  int pad[4];
};
struct AssocResult { double err; int best; int n_matches; };
struct AssocArgs {
  int n_queries, n_landmarks;
  const AssocQuery* queries;
  const AssocLandmark* landmarks;
  AssocResult* results;
  double pose[7];
  double edge_asso_2ddist, edge_asso_planedist, edge_asso_proj, edge_asso_angle;
  int assoc_near_frames;
  const double* plane_est; int plane_ld;
};
hipError_t launch_assoc(const AssocArgs& a, hipStream_t st);
// reproj_to_newplane: point i projected onto the plane in slot[i] of the plane state (slot < 0: copied through)
hipError_t launch_reproject(int n, const int* slot, const float* pts, const double* plane_est, int plane_ld, float* out, hipStream_t st);

}  // namespace pps
