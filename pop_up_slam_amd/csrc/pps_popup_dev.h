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
  double* obs_meas; int n_obs;
};

hipError_t launch_refresh_measurements(const RefreshArgs& a, hipStream_t st);

}  // namespace pps
