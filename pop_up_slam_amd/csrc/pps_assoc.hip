// pps_assoc.hip -- plane data association on the device.
//
// Replaces the candidate loop of Mapper_mono::findClosestPlane (/root/reference/pop_planar_slam/src/
// Mapping.cpp:256-397): for one newly popped-up plane, every stored landmark is gated (ground/wall class,
// frame distance, normal angle, plane distance, 2-D end-point distance, 1-D overlap of the ground
// segments) and scored; the lowest score wins.  The landmark planes are read straight from the solver's
// state in HBM (the estimate the last solve left there), nothing is downloaded.
//
// The sequential loop's result is reproduced exactly, including its corner cases:
//   * ground query: the FIRST live ground landmark wins at once, score stays -1           (:277-281)
//   * `numMatches==1 || total < best`: the first candidate that passes the gates is taken even when its
//     score is NaN (acos of a dot product that rounds above 1), and a NaN best is never displaced (:374)
//   * ties keep the earlier landmark.
// fp64 plane algebra, fp32 segment algebra, as in the reference.
#include <hip/hip_runtime.h>

#include "pps_geom.h"
#include "pps_popup_dev.h"

namespace pps {
namespace {

constexpr int kAssocThreads = 256;
constexpr double kPi = 3.14159265358979323846;   // isam::PI, util.h:40

__device__ __forceinline__ double norm3(const double* p) { return sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); }
__device__ __forceinline__ float norm2f(float x, float y) { return sqrtf(x * x + y * y); }

// Mapping.cpp:112-126
__device__ __forceinline__ float point_proj_to_lineseg(float bx, float by, float ex, float ey, float qx, float qy) {
  const float length = norm2f(ex - bx, ey - by);
  if ((double)length < 0.001) return norm2f(qx - bx, qy - by);
  const float t = ((qx - bx) * (ex - bx) + (qy - by) * (ey - by)) / length / length;
  if (t > 1.0f) return 1.0f;
  if (t < 0.0f) return 0.0f;
  return t;
}

struct Cand {
  int first;       // lowest index that passed the gates (INT_MAX: none)
  double ferr;     // its score
  int best;        // lowest index among the minimal non-NaN scores (INT_MAX: none)
  double berr;
};

__device__ __forceinline__ void cand_merge(Cand& a, const Cand& b) {
  if (b.first < a.first) { a.first = b.first; a.ferr = b.ferr; }
  if (b.best != 0x7fffffff && (a.best == 0x7fffffff || b.berr < a.berr || (b.berr == a.berr && b.best < a.best))) {
    a.best = b.best; a.berr = b.berr;
  }
}

__global__ __launch_bounds__(kAssocThreads) void k_assoc(AssocArgs a) {
  const int q = blockIdx.x;
  const AssocQuery Q = a.queries[q];
  const int fpi = Q.frame_plane_indice;
  double cur_world[4], n_cur[3];
  plane_transform_from(Q.plane_local, a.pose, cur_world);            // :264
  {
    const double l = norm3(cur_world);
    for (int k = 0; k < 3; k++) n_cur[k] = cur_world[k] / l;
  }
  double R[9];
  quat_to_R(a.pose + 3, R);
  // current plane in the sensor frame: normal / d / point0 (isam_plane3d.h:148-171)
  double n_loc[3], x0_loc[3];
  {
    const double l = norm3(Q.plane_local);
    const double d = -Q.plane_local[3] / l;
    for (int k = 0; k < 3; k++) { n_loc[k] = Q.plane_local[k] / l; x0_loc[k] = d * n_loc[k]; }
  }

  Cand c{0x7fffffff, 0.0, 0x7fffffff, 0.0};
  int n_match = 0;
  for (int i = threadIdx.x; i < a.n_landmarks; i += kAssocThreads) {
    const AssocLandmark L = a.landmarks[i];
    if (L.deleted || L.plane_slot < 0) continue;                      // :275
    if (fpi == 0 && L.frame_plane_indice == 0) {                      // :277-281 (first one wins, see reduction)
      if (i < c.first) { c.first = i; c.ferr = -1.0; }
      continue;
    }
    if ((fpi == 0 && L.frame_plane_indice >= 1) || (fpi >= 1 && L.frame_plane_indice == 0)) continue;   // :282-284
    if (fpi == 0) continue;                                            // ground query scores nothing else
    if (Q.frame_seq_id - L.frame_seq_id > a.assoc_near_frames) continue;   // :295
    double ow[4], ol[4];
    for (int k = 0; k < 4; k++) ow[k] = a.plane_est[(size_t)k * a.plane_ld + L.plane_slot];
    plane_transform_to_raw(ow, a.pose, R, ol);                        // :291
    normalize4(ol);
    double n_old[3];
    {
      const double l = norm3(ow);
      for (int k = 0; k < 3; k++) n_old[k] = ow[k] / l;
    }
    const double angle = acos(n_cur[0] * n_old[0] + n_cur[1] * n_old[1] + n_cur[2] * n_old[2]) * 180.0 / kPi;   // :298
    if (angle > a.edge_asso_angle) continue;                          // :304
    double thre2d = a.edge_asso_2ddist, thre_cov = a.edge_asso_proj;
    if (angle < 25.0) {                                               // :309-315
      thre_cov = a.edge_asso_proj / 3;
      thre2d = a.edge_asso_2ddist * 1.5;
      if (angle <= 10.0) thre_cov = a.edge_asso_proj / 2;
    }
    double x0[3];
    {
      const double l = norm3(ol);
      const double d = -ol[3] / l;
      for (int k = 0; k < 3; k++) x0[k] = d * (ol[k] / l);
    }
    const double plane_dist = fabs(n_loc[0] * (x0[0] - x0_loc[0]) + n_loc[1] * (x0[1] - x0_loc[1]) + n_loc[2] * (x0[2] - x0_loc[2]));   // :318
    if (plane_dist > a.edge_asso_planedist) continue;                 // :324
    if (plane_dist < 1.5) thre_cov = thre_cov / 2;                    // :326
    float d2 = 0.f, d2c = 0.f;                                         // :330-346
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const float u = norm2f(Q.seg2d[2 * e] - L.seg2d[0], Q.seg2d[2 * e + 1] - L.seg2d[1]);
      const float v = norm2f(Q.seg2d[2 * e] - L.seg2d[2], Q.seg2d[2 * e + 1] - L.seg2d[3]);
      d2 += (u < v ? u : v);
    }
    d2 = d2 / 2;
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const float u = norm2f(Q.seg2d[0] - L.seg2d[2 * e], Q.seg2d[1] - L.seg2d[2 * e + 1]);
      const float v = norm2f(Q.seg2d[2] - L.seg2d[2 * e], Q.seg2d[3] - L.seg2d[2 * e + 1]);
      d2c += (u < v ? u : v);
    }
    d2c = d2c / 2;
    if ((double)d2 > thre2d || (double)d2c > thre2d) continue;        // :350
    const float o_bg = point_proj_to_lineseg(Q.seg3d[0], Q.seg3d[1], Q.seg3d[2], Q.seg3d[3], L.seg3d[0], L.seg3d[1]);   // :355-360
    const float o_ed = point_proj_to_lineseg(Q.seg3d[0], Q.seg3d[1], Q.seg3d[2], Q.seg3d[3], L.seg3d[2], L.seg3d[3]);
    const float cov_on = fabsf(o_bg - o_ed);
    const float n_bg = point_proj_to_lineseg(L.seg3d[0], L.seg3d[1], L.seg3d[2], L.seg3d[3], Q.seg3d[0], Q.seg3d[1]);
    const float n_ed = point_proj_to_lineseg(L.seg3d[0], L.seg3d[1], L.seg3d[2], L.seg3d[3], Q.seg3d[2], Q.seg3d[3]);
    const float cov_no = fabsf(n_bg - n_ed);
    if ((double)cov_on < thre_cov || (double)cov_no < thre_cov) continue;   // :364
    double total = angle / a.edge_asso_angle * 3 + (double)(1.0f - cov_on) + (double)(1.0f - cov_no);   // :368
    total += (double)(d2 > d2c ? d2 : d2c) / thre2d + plane_dist / 4;                                     // :369
    n_match++;
    if (i < c.first) { c.first = i; c.ferr = total; }
    if (total == total && (c.best == 0x7fffffff || total < c.berr)) { c.best = i; c.berr = total; }   // ascending i per thread
  }

  __shared__ Cand sc[kAssocThreads];
  __shared__ int sn[kAssocThreads];
  sc[threadIdx.x] = c;
  sn[threadIdx.x] = n_match;
  __syncthreads();
  for (int s = kAssocThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      cand_merge(sc[threadIdx.x], sc[threadIdx.x + s]);
      sn[threadIdx.x] += sn[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const Cand r = sc[0];
    AssocResult out;
    if (r.first == 0x7fffffff) { out.best = -1; out.err = -1.0; out.n_matches = 0; }
    else if (fpi == 0) { out.best = r.first; out.err = -1.0; out.n_matches = 1; }
    else if (r.ferr != r.ferr) { out.best = r.first; out.err = r.ferr; out.n_matches = sn[0]; }   // NaN first match sticks
    else { out.best = r.best; out.err = r.berr; out.n_matches = sn[0]; }
    a.results[q] = out;
  }
}

// Mapper_mono::reproj_to_newplane (src/Mapping.cpp:609-632): stored polygon vertices projected onto the optimised plane of
// their landmark, Plane3d::project_to_plane (src/isam_plane3d.h:173-178) in fp64 on the fp32 vertex, result cast to fp32.
__global__ __launch_bounds__(256) void k_reproject(int n, const int* __restrict__ slot, const float* __restrict__ pts,
                                                   const double* __restrict__ plane_est, int plane_ld, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int sl = slot[i];
  const double x = (double)pts[3 * i], y = (double)pts[3 * i + 1], z = (double)pts[3 * i + 2];
  if (sl < 0) { out[3 * i] = pts[3 * i]; out[3 * i + 1] = pts[3 * i + 1]; out[3 * i + 2] = pts[3 * i + 2]; return; }   // merged / unknown landmark: untouched
  double p[4];
  for (int k = 0; k < 4; k++) p[k] = plane_est[(size_t)k * plane_ld + sl];
  const double l = norm3(p);
  const double nx = p[0] / l, ny = p[1] / l, nz = p[2] / l, dd = -p[3] / l;
  const double s = (nx * x + ny * y + nz * z) - dd;
  out[3 * i] = (float)(x - nx * s); out[3 * i + 1] = (float)(y - ny * s); out[3 * i + 2] = (float)(z - nz * s);
}

}  // namespace

hipError_t launch_reproject(int n, const int* slot, const float* pts, const double* plane_est, int plane_ld, float* out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_reproject, dim3((n + 255) / 256), dim3(256), 0, st, n, slot, pts, plane_est, plane_ld, out);
  return hipGetLastError();
}

hipError_t launch_assoc(const AssocArgs& a, hipStream_t st) {
  if (a.n_queries <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_assoc, dim3(a.n_queries), dim3(kAssocThreads), 0, st, a);
  return hipGetLastError();
}

}  // namespace pps
