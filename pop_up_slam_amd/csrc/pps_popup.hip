// pps_popup.hip -- pop-up kernels (fp32, like the reference's pop_up_wall).
//
//   K5  segments -> plane equations   popup_plane::get_plane_equation / update_plane_equation_from_seg
//                                      (/root/reference/pop_up_wall/libs/popup_plane.cpp:551-603,654-705)
//   K6  pixels -> 3-D points / depth   generate_cloud + matrixToCloud (:807-863,925-985), get_depth_map_good (:866-921),
//                                      ray_plane_interact (libs/matrix_utils.cpp:189-193)
// K5 and K6 are fused into one launch: every workgroup re-derives the (<= 64) plane equations of the
// frame into LDS (a few hundred flops), then its threads classify one pixel each against the plane
// polygons (also in LDS) and intersect the pixel ray with the plane.  One 16-byte record per pixel is
// written with a single coalesced store; the bound is HBM (20 B/pixel algorithmic: 3 B BGR + 1 B label
// equivalent in, 16 B out).
// k_refresh_measurements is K5 alone, writing straight into the graph's edge-measurement array
// (Mapper_mono::update_plane_measurement, pop_planar_slam/src/Mapping.cpp:590-607).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pps.h"
#include "pps_popup_dev.h"
#include "pps_raster.h"

namespace pps {

constexpr int kMaxPlanes = 64;
constexpr int kPlaneData = 4 * (kMaxPlanes + 1) + 6 * kMaxPlanes + 2 * kMaxPlanes;    // floats: plane equations | ground segments | plane info
constexpr int kPlaneBlock = kPlaneData + 2;       // ... | the number of the run that wrote them (pps_popup_planes_wait polls it) | pad
constexpr int kMaxVerts = 512;
constexpr size_t kInSegOff = 512, kInPolyOff = kInSegOff + sizeof(float) * 4 * kMaxPlanes, kInBytes = kInPolyOff + sizeof(float) * 2 * kMaxVerts;
static_assert(sizeof(int) * (kMaxPlanes + 2) <= kInSegOff, "poly_off does not fit its part of the input block");

// one wall plane from a ground segment; exact operation order of the reference (and of the oracle)
__host__ __device__ __forceinline__ void seg_to_plane(const float* __restrict__ seg, const float* invK, const float* T,
                                             const float gs[4], float out[4], float* __restrict__ seg3d_world = nullptr,
                                             float* __restrict__ info = nullptr) {
  float Pw[2][3], zs[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const float u = seg[2 * e], v = seg[2 * e + 1];
    float ray[3];
#pragma unroll
    for (int i = 0; i < 3; i++) ray[i] = invK[i * 3 + 0] * u + invK[i * 3 + 1] * v + invK[i * 3 + 2] * 1.f;
    const float frac = -gs[3] / (gs[0] * ray[0] + gs[1] * ray[1] + gs[2] * ray[2]);
    const float Ps[4] = {frac * ray[0], frac * ray[1], frac * ray[2], 1.f};
    zs[e] = Ps[2];
    float Ph[4];
#pragma unroll
    for (int i = 0; i < 4; i++) Ph[i] = T[i * 4 + 0] * Ps[0] + T[i * 4 + 1] * Ps[1] + T[i * 4 + 2] * Ps[2] + T[i * 4 + 3] * Ps[3];
#pragma unroll
    for (int i = 0; i < 3; i++) Pw[e][i] = Ph[i] / Ph[3];
  }
  if (info) {
    // all_plane_dist_to_cam: camera footprint to the world ground segment (point_dist_lineseg, matrix_utils.cpp:290-304), and
    // whether both end points lie in front of the camera (popup_plane.cpp:616-627)
    const float bx = Pw[0][0], by = Pw[0][1], ex = Pw[1][0], ey = Pw[1][1], qx = T[3], qy = T[7];
    const float len = sqrtf((ex - bx) * (ex - bx) + (ey - by) * (ey - by));
    float dcam;
    if (len < 0.001) dcam = sqrtf((qx - bx) * (qx - bx) + (qy - by) * (qy - by));
    else {
      const float t = ((qx - bx) * (ex - bx) + (qy - by) * (ey - by)) / len / len;
      if (t < 0.0) dcam = sqrtf((qx - bx) * (qx - bx) + (qy - by) * (qy - by));
      else if (t > 1.0) dcam = sqrtf((qx - ex) * (qx - ex) + (qy - ey) * (qy - ey));
      else {
        const float px = bx + t * (ex - bx), py = by + t * (ey - by);
        dcam = sqrtf((qx - px) * (qx - px) + (qy - py) * (qy - py));
      }
    }
    info[0] = dcam;
    info[1] = (zs[0] > 0 && zs[1] > 0) ? 1.f : 0.f;
  }
  if (seg3d_world) {   // ground_seg3d_lines_world row, z forced to exact zero (popup_plane.cpp:569-578)
    seg3d_world[0] = Pw[0][0]; seg3d_world[1] = Pw[0][1]; seg3d_world[2] = 0.f;
    seg3d_world[3] = Pw[1][0]; seg3d_world[4] = Pw[1][1]; seg3d_world[5] = 0.f;
  }
  const float t1[3] = {Pw[1][0] - Pw[0][0], Pw[1][1] - Pw[0][1], Pw[1][2] - Pw[0][2]};
  const float t2[3] = {0.f, 0.f, -1.f};
  const float nw[3] = {t1[1] * t2[2] - t1[2] * t2[1], t1[2] * t2[0] - t1[0] * t2[2], t1[0] * t2[1] - t1[1] * t2[0]};
  const float dist = -(nw[0] * Pw[0][0] + nw[1] * Pw[0][1] + nw[2] * Pw[0][2]);
  const float pw[4] = {nw[0], nw[1], nw[2], dist};
#pragma unroll
  for (int k = 0; k < 4; k++) out[k] = T[0 * 4 + k] * pw[0] + T[1 * 4 + k] * pw[1] + T[2 * 4 + k] * pw[2] + T[3 * 4 + k] * pw[3];
}

__host__ __device__ __forceinline__ void ground_plane_sensor(const float* T, float gs[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) gs[k] = T[0 * 4 + k] * 0.f + T[1 * 4 + k] * 0.f + T[2 * 4 + k] * -1.f + T[3 * 4 + k] * 0.f;
}

__global__ __launch_bounds__(64) void k_popup_planes(const float* __restrict__ seg2d, int n, PopupParams prm,
                                                     float* __restrict__ planes_out) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j > n) return;
  float gs[4], pl[4];
  ground_plane_sensor(prm.T, gs);
  if (j == 0) { pl[0] = gs[0]; pl[1] = gs[1]; pl[2] = gs[2]; pl[3] = gs[3]; }
  else seg_to_plane(seg2d + 4 * (j - 1), prm.invK, prm.T, gs, pl);
#pragma unroll
  for (int k = 0; k < 4; k++) planes_out[4 * j + k] = pl[k];
}

// Row intervals of every plane polygon, once per frame: closed_polygons_homo_pts / cv::fillConvexPoly (pps_raster.h).
// One thread per (scaled row, polygon) derives the fill span and the outline runs of that row and merges them into
// disjoint column runs (usually one).  Layout: row_iv[row][s_off[p] + p + i] (stride nverts + nplanes), row_cnt[row][p],
// boxes[p] = boundingRect of the truncated polygon.  Rows are in scaled coordinates (image row / step).
constexpr int kRowMergeCap = 17;    // lists up to this length are merged, longer ones stay raw (still exact)

// LDS of the polygon set-up: the polygons as fillConvexPoly sees them
struct PolySetup {
  float poly[2 * kMaxVerts];
  int2 q[kMaxVerts];                  // vertices as the integer points fillConvexPoly sees (box coordinates)
  RasterLine line[kMaxVerts];         // outline edge ending at vertex v
  int off[kMaxPlanes + 2];
  int4 box[kMaxPlanes];               // boundingRect of the truncated polygon: x, y, width, height (scaled coordinates)
};

// every thread of a 256-thread workgroup calls this; ends with a barrier
__device__ __forceinline__ void popup_poly_setup(const PopupParams& prm, const float* __restrict__ polys, const int* __restrict__ poly_off, int nplanes,
                                                 PolySetup& L) {
  const int tid = threadIdx.x;
  for (int i = tid; i <= nplanes; i += 256) L.off[i] = poly_off[i];
  __syncthreads();
  const int nverts = L.off[nplanes];
  const int S = prm.step;                                       // 2 = downsample_poly
  for (int i = tid; i < 2 * nverts; i += 256) L.poly[i] = S == 2 ? polys[i] / 2 : polys[i];   // new_polys_close / 2 (:86-87)
  __syncthreads();
  // boundingRect of the truncated points (matrix_to_points + boundingRect, popup_plane.cpp:89-91)
  if (tid < nplanes) {
    const int v0 = L.off[tid], v1 = L.off[tid + 1];
    int x0 = 0, y0b = 0, x1 = -1, y1 = -1;
    for (int v = v0; v < v1; v++) {
      const int x = (int)L.poly[2 * v], y = (int)L.poly[2 * v + 1];
      if (v == v0) { x0 = x1 = x; y0b = y1 = y; }
      x0 = min(x0, x); x1 = max(x1, x); y0b = min(y0b, y); y1 = max(y1, y);
    }
    L.box[tid] = make_int4(x0, y0b, x1 - x0 + 1, y1 - y0b + 1);
  }
  __syncthreads();
  // polygon - box origin in fp32, truncated again (:92-96)
  for (int v = tid; v < nverts; v += 256) {
    int p = 0;
    while (L.off[p + 1] <= v) p++;
    const int4 bx = L.box[p];
    L.q[v] = make_int2((int)(L.poly[2 * v] - (float)bx.x), (int)(L.poly[2 * v + 1] - (float)bx.y));
  }
  __syncthreads();
  // outline: the edge that ends at vertex v starts at the previous vertex (the last one for the first)
  for (int v = tid; v < nverts; v += 256) {
    int p = 0;
    while (L.off[p + 1] <= v) p++;
    const int4 bx = L.box[p];
    const int u = v > L.off[p] ? v - 1 : L.off[p + 1] - 1;
    L.line[v] = raster_line(bx.z, bx.w, L.q[u].x, L.q[u].y, L.q[v].x, L.q[v].y);
  }
  __syncthreads();
}

// sub-interval i of polygon p in scaled row `row` (i = 0: the fill span, i >= 1: outline edge i - 1), as lo | hi << 16 in frame
// columns; 1u (lo = 1 > hi = 0) = empty
__device__ __forceinline__ unsigned int popup_raw_interval(const PolySetup& L, int row, int p, int i, int Ws) {
  const int v0 = L.off[p], npts = L.off[p + 1] - v0;
  const int4 bx = L.box[p];
  const int cy = row - bx.y;
  if (!(npts > 0 && cy >= 0 && cy < bx.w)) return 1u;
  int lo = 0, hi = -1;
  const bool hit = i == 0 ? raster_fill_row(L.q + v0, npts, bx.z, bx.w, cy, lo, hi) : raster_line_row(L.line[v0 + i - 1], cy, lo, hi);
  if (!hit) return 1u;
  lo = max(lo, 0) + bx.x; hi = min(hi, bx.z - 1) + bx.x;       // inside the box image, then frame columns (:104-113)
  lo = max(lo, 0); hi = min(hi, Ws - 1);
  return lo <= hi ? ((unsigned int)lo | ((unsigned int)hi << 16)) : 1u;
}

// n raw sub-intervals in iv (thread-private scratch, n <= kRowMergeCap) -> disjoint column runs in out; returns their number
__device__ __forceinline__ int popup_merge_intervals(unsigned int* __restrict__ iv, int n, unsigned int* __restrict__ out) {
  for (int a = 1; a < n; a++) {                                // insertion sort by lo; empty intervals merge away below
    const unsigned int key = iv[a];
    int b = a - 1;
    while (b >= 0 && (iv[b] & 0xffffu) > (key & 0xffffu)) { iv[b + 1] = iv[b]; b--; }
    iv[b + 1] = key;
  }
  int m = 0;
  unsigned int cur = 1u;
  for (int a = 0; a < n; a++) {
    const unsigned int v = iv[a], lo = v & 0xffffu, hi = v >> 16;
    if (lo > hi) continue;
    if (m > 0 && lo <= (cur >> 16) + 1u) { if (hi > (cur >> 16)) cur = (cur & 0xffffu) | (hi << 16); }
    else { if (m > 0) out[m - 1] = cur; cur = v; m++; }
  }
  if (m > 0) out[m - 1] = cur;
  return m;
}

// disjoint column runs of polygon p in scaled row `row` -> out[0 .. return value); scr: kRowMergeCap words of this thread
__device__ __forceinline__ int popup_row_item(const PolySetup& L, int row, int p, int Ws, unsigned int* __restrict__ scr, unsigned int* __restrict__ out) {
  const int v0 = L.off[p], npts = L.off[p + 1] - v0;
  const int4 bx = L.box[p];
  const int cy = row - bx.y;
  if (!(npts > 0 && cy >= 0 && cy < bx.w)) return 0;
  const int n = npts + 1;
  const bool merge = n <= kRowMergeCap;
  unsigned int* iv = merge ? scr : out;
  for (int i = 0; i < n; i++) iv[i] = popup_raw_interval(L, row, p, i, Ws);
  return merge ? popup_merge_intervals(iv, n, out) : n;
}

__global__ __launch_bounds__(256) void k_popup_rows(PopupParams prm, const float* __restrict__ polys, const int* __restrict__ poly_off,
                                                    int nplanes, unsigned int* __restrict__ row_iv, int* __restrict__ row_cnt,
                                                    int4* __restrict__ boxes) {
  __shared__ PolySetup L;
  __shared__ unsigned int s_scr[256 * kRowMergeCap];
  const int tid = threadIdx.x;
  popup_poly_setup(prm, polys, poly_off, nplanes, L);
  if (blockIdx.x == 0 && tid < nplanes) boxes[tid] = L.box[tid];
  const int nverts = L.off[nplanes];
  const int S = prm.step;
  const int Ws = (prm.width + S - 1) / S, Hs = (prm.height + S - 1) / S;
  const int item = blockIdx.x * 256 + tid;
  if (item >= Hs * nplanes) return;
  const int row = item / nplanes, p = item - row * nplanes;
  const int E = nverts + nplanes;
  row_cnt[(size_t)row * nplanes + p] = popup_row_item(L, row, p, Ws, s_scr + tid * kRowMergeCap, row_iv + (size_t)row * E + L.off[p] + p);
}

// Fused K5 + K6.  grid = (ceil(W / 256), ceil(H / PX)): a thread owns one column of PX consecutive rows and compares its
// column with the row intervals k_popup_rows derived (loaded into LDS for the PX rows of the workgroup).
// FUSED (frames up to 640 x 480: BASELINE config 5): the workgroup derives the intervals of ITS rows itself -- polygon set-up and
// the (row, polygon) items of pps_raster.h in LDS -- instead of reading what a k_popup_rows launch in front wrote: one launch
// per frame.  On large frames every one of thousands of workgroups would repeat the set-up (1920 x 1080: 60.9 against 46.8 us),
// so those keep the two launches.
template <int PX, bool FUSED>
__global__ __launch_bounds__(256) void k_popup_frame(PopupParams prm, const float* __restrict__ seg2d, int n,
                                                     const float* __restrict__ polys, const int* __restrict__ poly_off, int nplanes,
                                                     const unsigned int* __restrict__ row_iv, const int* __restrict__ row_cnt,
                                                     const int4* __restrict__ boxes, const unsigned char* __restrict__ bgr,
                                                     float* __restrict__ planes_out, pps_point* __restrict__ cloud,
                                                     float* __restrict__ depth, int* __restrict__ plane_id,
                                                     unsigned int* __restrict__ n_valid, unsigned int run_seq) {
  __shared__ float s_planes[kMaxPlanes + 1][4];
  __shared__ int s_off[kMaxPlanes + 2];
  __shared__ int4 s_box[kMaxPlanes];              // boundingRect of the truncated polygon: x, y, width, height
  __shared__ unsigned int s_iv[(kMaxVerts + kMaxPlanes) * PX];   // per row, per polygon: lo | hi << 16, scaled image columns
  __shared__ int s_ivcnt[kMaxPlanes * PX];
  __shared__ float s_ceil[4];
  __shared__ unsigned int s_cnt;
  const int tid = threadIdx.x;
  const int S = prm.step;
  const int W = prm.width, H = prm.height;
  const int y0 = blockIdx.y * PX;
  if (!FUSED) {
    for (int i = tid; i <= nplanes; i += 256) s_off[i] = poly_off[i];
    if (tid < nplanes) s_box[tid] = boxes[tid];
  }
  // ---- K5: plane equations of this frame (every workgroup; block (0, 0) publishes them -- straight into the pinned host block the
  // caller reads, like the per-workgroup point counts: posted writes, no copy engine hop behind the kernel) ----
  const bool publish = blockIdx.x == 0 && blockIdx.y == 0;
  if (tid <= n && tid <= kMaxPlanes) {
    float gs[4], pl[4];
    ground_plane_sensor(prm.T, gs);
    if (tid == 0) { pl[0] = gs[0]; pl[1] = gs[1]; pl[2] = gs[2]; pl[3] = gs[3]; }
    else seg_to_plane(seg2d + 4 * (tid - 1), prm.invK, prm.T, gs, pl,
                      (publish && planes_out) ? planes_out + 4 * (kMaxPlanes + 1) + 6 * (tid - 1) : nullptr,
                      (publish && planes_out) ? planes_out + 4 * (kMaxPlanes + 1) + 6 * kMaxPlanes + 2 * (tid - 1) : nullptr);
#pragma unroll
    for (int k = 0; k < 4; k++) s_planes[tid][k] = pl[k];
    if (publish && planes_out) {
#pragma unroll
      for (int k = 0; k < 4; k++) planes_out[4 * tid + k] = pl[k];
    }
  }
  if (tid == 0) {
    // ceiling_plane_world = (0,0,-1,ceiling) ; sensor = T^T * world  (popup_plane.cpp:602-603)
#pragma unroll
    for (int k = 0; k < 4; k++)
      s_ceil[k] = prm.T[0 * 4 + k] * 0.f + prm.T[1 * 4 + k] * 0.f + prm.T[2 * 4 + k] * -1.f + prm.T[3 * 4 + k] * prm.ceiling_thre;
    s_cnt = 0;
  }
  __syncthreads();
  // (the plane equations are what the caller's graph construction waits for: block (0, 0) has just written them into the pinned host block --
  // the run's number goes behind them with system-scope release, so that pps_popup_planes_wait sees them microseconds after the launch
  // started, long before the pixels are done)
  if (publish && planes_out && tid == 0)
    __hip_atomic_store(reinterpret_cast<unsigned int*>(planes_out + kPlaneData), run_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  int E;                                                        // entries per row of the interval lists
  if (FUSED) {
    __shared__ PolySetup L;
    __shared__ unsigned int s_scr[kMaxPlanes * PX * kRowMergeCap];
    popup_poly_setup(prm, polys, poly_off, nplanes, L);
    for (int i = tid; i <= nplanes; i += 256) s_off[i] = L.off[i];
    if (tid < nplanes) s_box[tid] = L.box[tid];
    E = L.off[nplanes] + nplanes;
    const int Ws = (W + S - 1) / S;
    // one thread per sub-interval (the fill span of a row is the long one: it replays fillConvexPoly's edge events), raw
    // values straight into the row's list; then one thread per (row, polygon) sorts and merges its list in place
    for (int e = tid; e < E * PX; e += 256) {
      const int k = e / E, r = e - k * E, Y = y0 + k;
      int p = 0;
      while (L.off[p + 1] + p + 1 <= r) p++;
      s_iv[e] = (Y < H && Y % S == 0) ? popup_raw_interval(L, Y / S, p, r - L.off[p] - p, Ws) : 1u;
    }
    __syncthreads();
    for (int i = tid; i < nplanes * PX; i += 256) {             // (nplanes * PX <= 128 items, one thread each)
      const int k = i / nplanes, p = i - k * nplanes, Y = y0 + k;
      const int npts = L.off[p + 1] - L.off[p], n = npts + 1;
      const int cy = Y / S - L.box[p].y;
      int cnt = 0;
      if (Y < H && Y % S == 0 && npts > 0 && cy >= 0 && cy < L.box[p].w) {
        unsigned int* lst = s_iv + k * E + L.off[p] + p;
        cnt = n;
        if (n <= kRowMergeCap) {
          unsigned int* scr = s_scr + i * kRowMergeCap;
          for (int a = 0; a < n; a++) scr[a] = lst[a];
          cnt = popup_merge_intervals(scr, n, lst);
        }
      }
      s_ivcnt[k * nplanes + p] = cnt;
    }
  } else {
    E = s_off[nplanes] + nplanes;
    // interval lists of this workgroup's rows (rows that are not a multiple of the step have none)
    for (int i = tid; i < nplanes * PX; i += 256) {
      const int k = i / nplanes, p = i - k * nplanes, Y = y0 + k;
      s_ivcnt[k * nplanes + p] = (Y < H && Y % S == 0) ? row_cnt[(size_t)(Y / S) * nplanes + p] : 0;
    }
    for (int i = tid; i < E * PX; i += 256) {
      const int k = i / E, Y = y0 + k;
      s_iv[i] = (Y < H && Y % S == 0) ? row_iv[(size_t)(Y / S) * E + (i - k * E)] : 1u;
    }
  }
  __syncthreads();

  unsigned int kept = 0;
  // Column strip: the thread owns one x and PX consecutive rows (blockIdx.y); the rows are the same for all lanes, so the
  // interval reads are LDS broadcasts.
  const int x = blockIdx.x * 256 + tid;
  const bool xin = x < W;
  const float fx = (float)x;
  const bool xsel = x % S == 0;
  const unsigned int xs = (unsigned int)(x / S);
  int pid[PX];
#pragma unroll
  for (int k = 0; k < PX; k++) pid[k] = -1;
  if (xin && xsel) {
    for (int p = nplanes - 1; p >= 0; p--) {                 // a later plane overwrites an earlier one: scan backwards
      const int4 bx = s_box[p];
      if ((int)xs < bx.x || (int)xs >= bx.x + bx.z) continue;
      const unsigned int* iv = s_iv + s_off[p] + p;
      bool all_set = true;
#pragma unroll
      for (int k = 0; k < PX; k++) {
        if (pid[k] < 0) {
          const int cnt = s_ivcnt[k * nplanes + p];
          const unsigned int* r = iv + k * E;
          bool in = false;
          for (int a = 0; a < cnt; a++) { const unsigned int v = r[a]; in = in || (xs >= (v & 0xffffu) && xs <= (v >> 16)); }
          if (in) pid[k] = p;
        }
        all_set = all_set && pid[k] >= 0;
      }
      if (all_set) break;
    }
  }
#pragma unroll
  for (int k = 0; k < PX; k++) {
    const int y = y0 + k;
    bool keep = false;
    if (xin && y < H) {
      const int idx = y * W + x;
      const float fy = (float)y;
      const int pd = pid[k];        // rows that are not a multiple of the step hold no intervals
      pps_point pt;
      pt.x = pt.y = pt.z = 0.f;
      pt.rgba = 0u;
      float dep = 0.f;
      if (pd >= 0) {
        // ---- K6: ray-plane intersection (ray_plane_interact), world transform, filters ----
        const float* pl = s_planes[pd];
        float ray[3];
#pragma unroll
        for (int i = 0; i < 3; i++) ray[i] = prm.invK[i * 3 + 0] * fx + prm.invK[i * 3 + 1] * fy + prm.invK[i * 3 + 2] * 1.f;
        const float frac = -pl[3] / (pl[0] * ray[0] + pl[1] * ray[1] + pl[2] * ray[2]);
        const float Ps[3] = {frac * ray[0], frac * ray[1], frac * ray[2]};
        float Pw[3];
#pragma unroll
        for (int i = 0; i < 3; i++) Pw[i] = prm.T[i * 4 + 0] * Ps[0] + prm.T[i * 4 + 1] * Ps[1] + prm.T[i * 4 + 2] * Ps[2];
#pragma unroll
        for (int i = 0; i < 3; i++) Pw[i] += prm.T[i * 4 + 3];
        keep = !(Ps[2] < 0.f) && !(Ps[2] > prm.depth_thre) && !(Pw[2] < -0.2f);
        if (keep) {
          pt.x = Pw[0]; pt.y = Pw[1];
          pt.z = Pw[2] < prm.ceiling_thre ? Pw[2] : prm.ceiling_thre;
          unsigned int rgb = 0u;
          if (bgr) {
            const unsigned char* c = bgr + 3 * (size_t)idx;
            rgb = ((unsigned int)c[2] << 16) | ((unsigned int)c[1] << 8) | (unsigned int)c[0];
          }
          pt.rgba = (1u << 24) | rgb;
        }
        // depth map (get_depth_map_good): ceiling plane substituted above the ceiling threshold
        if (Pw[2] < prm.ceiling_thre) {
          if (!(Ps[2] < 0.f)) dep = Ps[2];
        } else {
          const float fc = -s_ceil[3] / (s_ceil[0] * ray[0] + s_ceil[1] * ray[1] + s_ceil[2] * ray[2]);
          const float z = fc * ray[2];
          if (!(z < 0.f)) dep = z;
        }
      }
      cloud[idx] = pt;                     // one 16-byte store per lane, a row segment per wave
      if (depth) depth[idx] = dep;
      if (plane_id) plane_id[idx] = pd;
    }
    kept += (unsigned int)__popcll(__ballot(keep));
  }
  if ((tid & 63) == 0 && kept) atomicAdd(&s_cnt, kept);
  __syncthreads();
  // one count per workgroup, summed on the host: hundreds of workgroups finish together, and as many atomics on one word queue up at the L2
  if (tid == 0) n_valid[blockIdx.y * gridDim.x + blockIdx.x] = s_cnt;
}

// K5 feeding the graph: one thread per (frame, plane).  Pose comes from the fp64 estimate, is cast to
// fp32 like `value().wTo().cast<float>()` (Mapping.cpp:598-599); the fp32 plane is cast back to fp64 and
// normalised like Plane3d(Vector4d) (src/isam_plane3d.h:59-66) before it lands in the edge array.
__global__ __launch_bounds__(256) void k_refresh_measurements(RefreshArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_items) return;
  const int slot = a.item_slot[i];
  if (slot < 0) return;
  const int f = a.item_frame[i], j = a.item_plane[i];
  const int ps = a.frame_pose_slot[f];
  double q[4], t[3];
  t[0] = a.pose_est[(size_t)0 * a.pose_ld + ps]; t[1] = a.pose_est[(size_t)1 * a.pose_ld + ps]; t[2] = a.pose_est[(size_t)2 * a.pose_ld + ps];
  q[0] = a.pose_est[(size_t)3 * a.pose_ld + ps]; q[1] = a.pose_est[(size_t)4 * a.pose_ld + ps];
  q[2] = a.pose_est[(size_t)5 * a.pose_ld + ps]; q[3] = a.pose_est[(size_t)6 * a.pose_ld + ps];
  // wTo in fp64 (Pose3d::wTo, Pose3d.h:188-194), then cast
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  float T[16];
  T[0] = (float)(1 - (tyy + tzz)); T[1] = (float)(txy - twz);       T[2] = (float)(txz + twy);        T[3] = (float)t[0];
  T[4] = (float)(txy + twz);       T[5] = (float)(1 - (txx + tzz)); T[6] = (float)(tyz - twx);        T[7] = (float)t[1];
  T[8] = (float)(txz - twy);       T[9] = (float)(tyz + twx);       T[10] = (float)(1 - (txx + tyy)); T[11] = (float)t[2];
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
  float gs[4], pl[4];
  ground_plane_sensor(T, gs);
  if (j == 0) { pl[0] = gs[0]; pl[1] = gs[1]; pl[2] = gs[2]; pl[3] = gs[3]; }
  else seg_to_plane(a.seg2d + 4 * (size_t)(a.frame_seg_off[f] + j - 1), a.invK, T, gs, pl);
  double v[4] = {(double)pl[0], (double)pl[1], (double)pl[2], (double)pl[3]};
  const double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  // A ground ray that no longer hits the ground (segment at or above the horizon of the refreshed pose)
  // gives inf/NaN; the reference would push that NaN into the factor (it only logs NaN planes at
  // creation, main_3d.cpp:434-435) and lose the graph.  Keep the previous measurement instead.
  if (!(nrm > 0.0) || !(nrm < 1e300)) return;
#pragma unroll
  for (int k = 0; k < 4; k++) a.obs_meas[(size_t)k * a.obs_ld + slot] = v[k] / nrm;
}

hipError_t launch_refresh_measurements(const RefreshArgs& a, hipStream_t st) {
  if (a.n_items == 0) return hipSuccess;
  hipLaunchKernelGGL(k_refresh_measurements, dim3((a.n_items + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace pps

// ============================================================================================
using namespace pps;

struct pps_popup {
  int device = 0, width = 0, height = 0;
  float invK[9];
  std::string err;
  hipStream_t stream = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  unsigned char* d_bgr = nullptr; bool has_image = false;
  pps_point* d_cloud = nullptr;
  float* d_depth = nullptr;
  float* d_depth_fill = nullptr;   // second depth buffer of pps_popup_fill_depth (allocated on first use)
  int last_step = 1;
  float last_T[16] = {0};          // pose of the last run
  int* d_pid = nullptr;
  bool want_depth = true, want_pid = true;   // optional per-pixel outputs (pps_popup_set_outputs)
  float* d_planes = nullptr;   // (kMaxPlanes+1) x 4 plane equations, then kMaxPlanes x 6 world ground segments, kMaxPlanes x 2 plane info; d_count behind
  float* d_seg = nullptr;      // kMaxPlanes x 4
  float* d_polys = nullptr;    // 2*kMaxVerts
  int* d_off = nullptr;        // kMaxPlanes+2
  char* d_in = nullptr;        // the block d_off / d_seg / d_polys point into
  char* h_in = nullptr;        // pinned staging of the same layout
  unsigned int* d_row_iv = nullptr;   // row intervals of the last run (k_popup_rows): height x (kMaxVerts + kMaxPlanes)
  int* d_row_cnt = nullptr;           // height x kMaxPlanes
  int4* d_boxes = nullptr;            // kMaxPlanes
  unsigned int* d_count = nullptr;   // kept points per workgroup of the last run
  size_t count_cap = 0;
  unsigned int run_seq = 0;          // number of the last run (the frame kernel publishes it behind the plane equations)
  bool in_flight = false;            // pps_popup_run_async was not waited for yet
  size_t n_wg_last = 0;              // workgroups of the last run (their point counts sit in h_count)
  unsigned int* h_count = nullptr;   // pinned (behind h_planes)
  float* h_planes = nullptr;         // pinned mirror of d_planes as the last run left it: [plane equations | ground segments | plane info]
  int last_n = 0;
  double last_kernel_s = 0;
};

namespace {
int pfail(pps_popup* p, int code, const std::string& m) { if (p) p->err = m; return code; }
#define PHIP(p, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return pfail(p, PPS_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)
}  // namespace

extern "C" {

int pps_popup_planes(int device, const float* seg2d, int n, const float invK[9], const float T_wc[16], float* planes_out) {
  if (!seg2d || !invK || !T_wc || !planes_out || n < 0) return PPS_EINVAL;
  if (hipSetDevice(device) != hipSuccess) return PPS_EHIP;
  float *d_seg = nullptr, *d_out = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d_seg), sizeof(float) * 4 * (size_t)(n > 0 ? n : 1)) != hipSuccess) return PPS_EHIP;
  if (hipMalloc(reinterpret_cast<void**>(&d_out), sizeof(float) * 4 * (size_t)(n + 1)) != hipSuccess) { (void)hipFree(d_seg); return PPS_EHIP; }
  PopupParams prm{};
  memcpy(prm.invK, invK, sizeof prm.invK);
  memcpy(prm.T, T_wc, sizeof prm.T);
  hipError_t e = hipSuccess;
  if (n > 0) e = hipMemcpy(d_seg, seg2d, sizeof(float) * 4 * (size_t)n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_popup_planes, dim3((n + 1 + 63) / 64), dim3(64), 0, 0, d_seg, n, prm, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(planes_out, d_out, sizeof(float) * 4 * (size_t)(n + 1), hipMemcpyDeviceToHost);
  (void)hipFree(d_seg);
  (void)hipFree(d_out);
  return e == hipSuccess ? PPS_OK : PPS_EHIP;
}

int pps_popup_create(int device, int width, int height, const float invK[9], pps_popup** out) {
  if (!out || !invK || width <= 0 || height <= 0 || width > 32768 || height > 32768) return PPS_EINVAL;
  pps_popup* p = new (std::nothrow) pps_popup();
  if (!p) return PPS_ENOMEM;
  p->device = device; p->width = width; p->height = height;
  memcpy(p->invK, invK, sizeof p->invK);
  const size_t npx = (size_t)width * height;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&p->ev[0]);
  if (e == hipSuccess) e = hipEventCreate(&p->ev[1]);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_bgr), npx * 3);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_cloud), npx * sizeof(pps_point));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_depth), npx * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_pid), npx * sizeof(int));
  // per-frame inputs in one block (one transfer per frame): [poly_off | seg2d | polygons]
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_in), kInBytes);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&p->h_in), kInBytes, hipHostMallocDefault);
  if (e == hipSuccess) {
    p->d_off = reinterpret_cast<int*>(p->d_in);
    p->d_seg = reinterpret_cast<float*>(p->d_in + kInSegOff);
    p->d_polys = reinterpret_cast<float*>(p->d_in + kInPolyOff);
  }
  p->count_cap = (size_t)((width + 255) / 256) * (size_t)((height + 1) / 2);      // workgroups of the finest launch geometry (2 rows each)
  // small results in one block (one transfer per frame, arriving with the run's synchronisation): [plane equations | ground
  // segments | plane info | kept points per workgroup]
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_planes), sizeof(float) * kPlaneBlock + sizeof(unsigned int) * p->count_cap);
  if (e == hipSuccess) p->d_count = reinterpret_cast<unsigned int*>(p->d_planes + kPlaneBlock);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_row_iv), sizeof(unsigned int) * (size_t)height * (kMaxVerts + kMaxPlanes));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_row_cnt), sizeof(int) * (size_t)height * kMaxPlanes);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_boxes), sizeof(int4) * kMaxPlanes);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&p->h_planes), sizeof(float) * kPlaneBlock + sizeof(unsigned int) * p->count_cap, hipHostMallocDefault);
  if (e == hipSuccess) p->h_count = reinterpret_cast<unsigned int*>(p->h_planes + kPlaneBlock);
  if (e != hipSuccess) { pps_popup_destroy(p); return PPS_EHIP; }
  *out = p;
  return PPS_OK;
}

int pps_popup_destroy(pps_popup* p) {
  if (!p) return PPS_EINVAL;
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  (void)hipFree(p->d_bgr); (void)hipFree(p->d_cloud); (void)hipFree(p->d_depth); (void)hipFree(p->d_depth_fill); (void)hipFree(p->d_pid);
  (void)hipFree(p->d_planes); (void)hipFree(p->d_in); if (p->h_in) (void)hipHostFree(p->h_in);
  (void)hipFree(p->d_row_iv); (void)hipFree(p->d_row_cnt); (void)hipFree(p->d_boxes);
  if (p->h_planes) (void)hipHostFree(p->h_planes);
  if (p->ev[0]) (void)hipEventDestroy(p->ev[0]);
  if (p->ev[1]) (void)hipEventDestroy(p->ev[1]);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return PPS_OK;
}

const char* pps_popup_last_error(const pps_popup* p) { return p ? p->err.c_str() : "null handle"; }

int pps_popup_set_image(pps_popup* p, const unsigned char* bgr) {
  if (!p) return PPS_EINVAL;
  if (!bgr) { p->has_image = false; return PPS_OK; }
  PHIP(p, hipSetDevice(p->device));
  PHIP(p, hipMemcpy(p->d_bgr, bgr, (size_t)p->width * p->height * 3, hipMemcpyHostToDevice));
  p->has_image = true;
  return PPS_OK;
}

// a run enqueued by pps_popup_run_async is over before anything reads its results
static int popup_settle(pps_popup* p) {
  if (p && p->in_flight) { PHIP(p, hipSetDevice(p->device)); PHIP(p, hipStreamSynchronize(p->stream)); p->in_flight = false; }
  return PPS_OK;
}

// one run enqueued on the handle's stream; timed: with the event pair around the kernel (the synchronous entry point)
static int popup_enqueue(pps_popup* p, const float* seg2d, int n, const float T_wc[16], const float* polys, const int* poly_off, int nplanes, int step,
                         float depth_thre, float ceiling_thre, bool timed) {
  if (!p || !T_wc || !poly_off || n < 0 || nplanes < 0) return PPS_EINVAL;
  if (n > kMaxPlanes - 1 || nplanes > kMaxPlanes) return pfail(p, PPS_EINVAL, "too many planes for one frame (max 64)");
  if (nplanes > n + 1) return pfail(p, PPS_EINVAL, "more polygons than planes");
  if (poly_off[nplanes] > kMaxVerts) return pfail(p, PPS_EINVAL, "too many polygon vertices (max 512)");
  if (step != 1 && step != 2) return pfail(p, PPS_EINVAL, "step must be 1 or 2");
  PHIP(p, hipSetDevice(p->device));
  if ((n > 0 && !seg2d) || (poly_off[nplanes] > 0 && !polys)) return PPS_EINVAL;
  if (p->in_flight) { PHIP(p, hipStreamSynchronize(p->stream)); p->in_flight = false; }      // (the pinned input block is reused)
  PopupParams prm{};
  memcpy(prm.invK, p->invK, sizeof prm.invK);
  memcpy(prm.T, T_wc, sizeof prm.T);
  prm.width = p->width; prm.height = p->height; prm.step = step;
  prm.depth_thre = depth_thre; prm.ceiling_thre = ceiling_thre;
  // (the previous run ended with a stream sync: the pinned block is free)
  memcpy(p->h_in, poly_off, sizeof(int) * (size_t)(nplanes + 1));
  if (n > 0) memcpy(p->h_in + kInSegOff, seg2d, sizeof(float) * 4 * (size_t)n);
  const size_t poly_bytes = sizeof(float) * 2 * (size_t)poly_off[nplanes];
  if (poly_bytes > 0) memcpy(p->h_in + kInPolyOff, polys, poly_bytes);
  const int npx = p->width * p->height;
  PHIP(p, hipMemcpyAsync(p->d_in, p->h_in, kInPolyOff + poly_bytes, hipMemcpyHostToDevice, p->stream));
  if (timed) PHIP(p, hipEventRecord(p->ev[0], p->stream));
  p->run_seq++;
  // column strips of 256 x PX pixels: 2 rows per thread on small frames (640x480: 720 workgroups), 8 from ~1 Mpixel up;
  // frames up to 640 x 480 derive their row intervals inside the frame kernel (one launch), larger ones in a launch of their own
  const int pxt = npx >= (1 << 20) ? 8 : 2;
  const bool fused = npx <= 640 * 480 && nplanes > 0;
  const dim3 grid((p->width + 255) / 256, (p->height + pxt - 1) / pxt);
  const unsigned char* img = p->has_image ? p->d_bgr : nullptr;
  float* dep = p->want_depth ? p->d_depth : nullptr;
  int* pidp = p->want_pid ? p->d_pid : nullptr;
  if (nplanes > 0 && !fused) {
    const int hs = (p->height + step - 1) / step;
    hipLaunchKernelGGL(k_popup_rows, dim3((hs * nplanes + 255) / 256), dim3(256), 0, p->stream, prm, p->d_polys, p->d_off, nplanes,
                       p->d_row_iv, p->d_row_cnt, p->d_boxes);
  }
  if (fused)
    hipLaunchKernelGGL((k_popup_frame<2, true>), grid, dim3(256), 0, p->stream, prm, p->d_seg, n, p->d_polys, p->d_off, nplanes, p->d_row_iv, p->d_row_cnt,
                       p->d_boxes, img, p->h_planes, p->d_cloud, dep, pidp, p->h_count, p->run_seq);
  else if (pxt == 8)
    hipLaunchKernelGGL((k_popup_frame<8, false>), grid, dim3(256), 0, p->stream, prm, p->d_seg, n, p->d_polys, p->d_off, nplanes, p->d_row_iv, p->d_row_cnt,
                       p->d_boxes, img, p->h_planes, p->d_cloud, dep, pidp, p->h_count, p->run_seq);
  else
    hipLaunchKernelGGL((k_popup_frame<2, false>), grid, dim3(256), 0, p->stream, prm, p->d_seg, n, p->d_polys, p->d_off, nplanes, p->d_row_iv, p->d_row_cnt,
                       p->d_boxes, img, p->h_planes, p->d_cloud, dep, pidp, p->h_count, p->run_seq);
  PHIP(p, hipGetLastError());
  if (timed) PHIP(p, hipEventRecord(p->ev[1], p->stream));
  p->n_wg_last = (size_t)grid.x * grid.y;
  p->last_n = n; p->last_step = step;
  memcpy(p->last_T, T_wc, sizeof p->last_T);
  p->in_flight = true;
  return PPS_OK;
}

int pps_popup_run(pps_popup* p, const float* seg2d, int n, const float T_wc[16], const float* polys, const int* poly_off,
                  int nplanes, int step, float depth_thre, float ceiling_thre, int* n_valid) {
  const int rc = popup_enqueue(p, seg2d, n, T_wc, polys, poly_off, nplanes, step, depth_thre, ceiling_thre, true);
  if (rc != PPS_OK) return rc;
  // (plane block and point counts were written into pinned host memory by the kernel itself)
  PHIP(p, hipStreamSynchronize(p->stream));
  p->in_flight = false;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, p->ev[0], p->ev[1]);
  p->last_kernel_s = 1e-3 * ms;
  if (n_valid) { unsigned int tot = 0; for (size_t i = 0; i < p->n_wg_last; i++) tot += p->h_count[i]; *n_valid = (int)tot; }
  return PPS_OK;
}

// The same run without waiting for it (round 6: the frame loop's graph construction needs the plane equations, which the kernel's first
// workgroup publishes at once; the pixels -- cloud, depth, plane ids -- are nobody's input before the frame is drawn).
int pps_popup_run_async(pps_popup* p, const float* seg2d, int n, const float T_wc[16], const float* polys, const int* poly_off,
                        int nplanes, int step, float depth_thre, float ceiling_thre) {
  return popup_enqueue(p, seg2d, n, T_wc, polys, poly_off, nplanes, step, depth_thre, ceiling_thre, false);
}
// the plane equations of the run in flight ((n + 1) x 4), as soon as its first workgroup has written them
int pps_popup_planes_wait(pps_popup* p, float* planes) {
  if (!p || !planes) return PPS_EINVAL;
  if (p->run_seq == 0) return pfail(p, PPS_ESTATE, "no pop-up run");
  const volatile unsigned int* seq = reinterpret_cast<const volatile unsigned int*>(p->h_planes + kPlaneData);
  if (p->in_flight) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (*seq != p->run_seq) {
      if ((++spins & 0x3ff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.5) {
        PHIP(p, hipStreamSynchronize(p->stream));
        p->in_flight = false;
        if (*seq != p->run_seq) return pfail(p, PPS_EHIP, "the pop-up kernel did not publish its plane equations");
        break;
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  memcpy(planes, p->h_planes, sizeof(float) * 4 * (size_t)(p->last_n + 1));
  return PPS_OK;
}
// the end of the run in flight: its pixels are on the device (pps_popup_download), n_valid = points of its cloud
int pps_popup_wait(pps_popup* p, int* n_valid) {
  if (!p) return PPS_EINVAL;
  if (p->in_flight) { PHIP(p, hipSetDevice(p->device)); PHIP(p, hipStreamSynchronize(p->stream)); p->in_flight = false; }
  if (n_valid) { unsigned int tot = 0; for (size_t i = 0; i < p->n_wg_last; i++) tot += p->h_count[i]; *n_valid = (int)tot; }
  return PPS_OK;
}

// Tail of get_depth_map_good for the half-resolution pop-up (popup_plane.cpp:913-917): the depth map holds values on the
// even pixels only; cv::resize(0.5) -- INTER_AREA for an exact factor 2: the mean of a 2x2 block, three of whose pixels
// are 0 -- times 4 gives back the value at (2x, 2y); cv::resize(2, INTER_LINEAR) then spreads the half-size map over the
// full frame: source coordinate (X + 0.5) / 2 - 0.5, clamped at the borders, weights 0.25 / 0.75.
__global__ __launch_bounds__(256) void k_depth_fill(const float* __restrict__ sparse, float* __restrict__ out, int w, int h) {
  const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
  if (X >= w) return;
  const int hw = w / 2, hh = h / 2;
  float fx = (float)((X + 0.5) * 0.5 - 0.5), fy = (float)((Y + 0.5) * 0.5 - 0.5);
  int sx = (int)floorf(fx), sy = (int)floorf(fy);
  fx -= sx; fy -= sy;
  if (sx < 0) { fx = 0; sx = 0; }
  if (sx >= hw - 1) { fx = 0; sx = hw - 1; }
  if (sy < 0) { fy = 0; sy = 0; }
  if (sy >= hh - 1) { fy = 0; sy = hh - 1; }
  const int sx1 = sx + 1 < hw ? sx + 1 : hw - 1, sy1 = sy + 1 < hh ? sy + 1 : hh - 1;
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
  auto S = [&](int y, int x) { return sparse[(size_t)(2 * y) * w + 2 * x] * 0.25f * 4.0f; };
  const float r0 = S(sy, sx) * a0 + S(sy, sx1) * a1;
  const float r1 = S(sy1, sx) * a0 + S(sy1, sx1) * a1;
  out[(size_t)Y * w + X] = r0 * b0 + r1 * b1;
}

int pps_popup_fill_depth(pps_popup* p) {
  { const int rc0 = popup_settle(p); if (rc0 != PPS_OK) return rc0; }
  if (!p) return PPS_EINVAL;
  if (!p->want_depth) return pfail(p, PPS_ESTATE, "depth output is switched off (pps_popup_set_outputs)");
  if (p->last_step != 2) return pfail(p, PPS_ESTATE, "pps_popup_fill_depth follows a pps_popup_run with step = 2");
  if ((p->width | p->height) & 1) return pfail(p, PPS_EINVAL, "pps_popup_fill_depth needs an even image size");
  PHIP(p, hipSetDevice(p->device));
  const size_t npx = (size_t)p->width * p->height;
  if (!p->d_depth_fill) PHIP(p, hipMalloc(reinterpret_cast<void**>(&p->d_depth_fill), npx * sizeof(float)));
  hipLaunchKernelGGL(k_depth_fill, dim3((p->width + 255) / 256, p->height), dim3(256), 0, p->stream, p->d_depth, p->d_depth_fill, p->width, p->height);
  PHIP(p, hipGetLastError());
  PHIP(p, hipStreamSynchronize(p->stream));
  std::swap(p->d_depth, p->d_depth_fill);
  p->last_step = 1;                       // the map is dense now
  return PPS_OK;
}

int pps_popup_download(pps_popup* p, float* planes, pps_point* cloud, float* depth, int32_t* plane_id) {
  { const int rc0 = popup_settle(p); if (rc0 != PPS_OK) return rc0; }
  if (!p) return PPS_EINVAL;
  PHIP(p, hipSetDevice(p->device));
  const size_t npx = (size_t)p->width * p->height;
  if (planes) memcpy(planes, p->h_planes, sizeof(float) * 4 * (size_t)(p->last_n + 1));     // (came back with the run)
  if (cloud) PHIP(p, hipMemcpy(cloud, p->d_cloud, npx * sizeof(pps_point), hipMemcpyDeviceToHost));
  if (depth && !p->want_depth) return pfail(p, PPS_ESTATE, "depth output is switched off (pps_popup_set_outputs)");
  if (plane_id && !p->want_pid) return pfail(p, PPS_ESTATE, "plane-id output is switched off (pps_popup_set_outputs)");
  if (depth) PHIP(p, hipMemcpy(depth, p->d_depth, npx * sizeof(float), hipMemcpyDeviceToHost));
  if (plane_id) PHIP(p, hipMemcpy(plane_id, p->d_pid, npx * sizeof(int), hipMemcpyDeviceToHost));
  return PPS_OK;
}

int pps_popup_set_outputs(pps_popup* p, int want_depth, int want_plane_id) {
  if (!p) return PPS_EINVAL;
  p->want_depth = want_depth != 0; p->want_pid = want_plane_id != 0;
  return PPS_OK;
}

int pps_popup_plane_info(pps_popup* p, float plane_cam_dist_thre, const int* actual_plane_indices, int n_actual, float* dist_to_cam,
                         int32_t* good) {
  { const int rc0 = popup_settle(p); if (rc0 != PPS_OK) return rc0; }
  if (!p || (!dist_to_cam && !good) || n_actual < 0 || (n_actual > 0 && !actual_plane_indices)) return PPS_EINVAL;
  PHIP(p, hipSetDevice(p->device));
  const int n = p->last_n;
  std::vector<float> info(2 * (size_t)(n > 0 ? n : 1));
  if (n > 0) memcpy(info.data(), p->h_planes + 4 * (kMaxPlanes + 1) + 6 * kMaxPlanes, sizeof(float) * 2 * (size_t)n);
  if (dist_to_cam) dist_to_cam[0] = p->last_T[11];                 // the ground: camera height (transToWolrd(2,3), :617)
  if (good) good[0] = 1;                                           // "always push ground plane" (:620)
  for (int sgi = 0; sgi < n; sgi++) {
    if (dist_to_cam) dist_to_cam[sgi + 1] = info[2 * sgi];
    if (!good) continue;
    bool ok = info[2 * sgi + 1] != 0.f && info[2 * sgi] < plane_cam_dist_thre;   // :624-627
    if (ok && n_actual > 0) {                                      // manually connected edges are not popped up (:629-633)
      ok = false;
      for (int k = 0; k < n_actual; k++) ok = ok || actual_plane_indices[k] == sgi + 1;
    }
    good[sgi + 1] = ok ? 1 : 0;
  }
  return PPS_OK;
}

int pps_popup_download_segments3d(pps_popup* p, float* seg3d_world) {
  { const int rc0 = popup_settle(p); if (rc0 != PPS_OK) return rc0; }
  if (!p || !seg3d_world) return PPS_EINVAL;
  PHIP(p, hipSetDevice(p->device));
  if (p->last_n > 0)
    memcpy(seg3d_world, p->h_planes + 4 * (kMaxPlanes + 1), sizeof(float) * 6 * (size_t)p->last_n);
  return PPS_OK;
}

// popup_plane::find_2d_3d_closed_polygon_simplemode (libs/popup_plane.cpp:409-500) with walllength_threshold <= 0 (the class
// default, popup_plane.h:81): the closed 2-D polygon of every wall -- ground segment, image-boundary hits of the world-vertical
// lines through its end points (direction_hit_boundary, libs/matrix_utils.cpp:229-270), image corners in between.  Sequential
// fp32 work on a handful of segments: host code, like the segment selection of pps_edges_select.  The world ground points come
// from the same seg_to_plane the kernels run.  (The wall-length cut of :502-546 needs cv::intersectConvexConvex: not offered.)
namespace {
void hit_boundary(const float pt[2], const float direc[2], int w, int h, float hit[2]) {
  float lambd;
  if (direc[1] < 0) {
    lambd = (float)((0.0 - pt[1]) / direc[1]);
    if (lambd >= 0) { const float hx = pt[0] + lambd * direc[0], hy = pt[1] + lambd * direc[1]; if ((0 <= (int)hx) && ((int)hx <= w - 1)) { hit[0] = hx; hit[1] = hy; return; } }
  }
  if (direc[1] > 0) {
    lambd = (float)((h - 1.0 - pt[1]) / direc[1]);
    if (lambd >= 0) { const float hx = pt[0] + lambd * direc[0], hy = pt[1] + lambd * direc[1]; if ((0 <= (int)hx) && ((int)hx <= w - 1)) { hit[0] = hx; hit[1] = hy; return; } }
  }
  if (direc[0] > 0) {
    lambd = (float)((w - 1.0 - pt[0]) / direc[0]);
    if (lambd >= 0) { const float hx = pt[0] + lambd * direc[0], hy = pt[1] + lambd * direc[1]; if ((0 <= (int)hy) && ((int)hy <= h - 1)) { hit[0] = hx; hit[1] = hy; return; } }
  }
  if (direc[0] < 0) {
    lambd = (float)((0.0 - pt[0]) / direc[0]);
    if (lambd >= 0) { const float hx = pt[0] + lambd * direc[0], hy = pt[1] + lambd * direc[1]; if ((0 <= (int)hy) && ((int)hy <= h - 1)) { hit[0] = hx; hit[1] = hy; return; } }
  }
  hit[0] = -1.f; hit[1] = -1.f;
}
}  // namespace

int pps_popup_polygons_simple(const float K[9], const float invK[9], const float T_wc[16], int width, int height, const float* seg2d, int n,
                              float* verts, int cap_verts, int* poly_off, int* n_verts) {
  if (!K || !invK || !T_wc || !poly_off || n < 0 || width <= 0 || height <= 0 || (n > 0 && (!seg2d || !verts))) return PPS_EINVAL;
  if (cap_verts < 8 * n) return PPS_EINVAL;
  const float* T = T_wc;
  poly_off[0] = 0; poly_off[1] = 0;                      // the ground has no polygon in this mode (:489)
  float gs[4];
  ground_plane_sensor(T, gs);
  float iT[12];                                          // rows 0..2 of the rigid inverse of T_wc
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) iT[i * 4 + j] = T[j * 4 + i];
    iT[i * 4 + 3] = -(T[0 * 4 + i] * T[3] + T[1 * 4 + i] * T[7] + T[2 * 4 + i] * T[11]);
  }
  int nv = 0;
  for (int sg = 0; sg < n; sg++) {
    float pl[4], s3[6];
    seg_to_plane(seg2d + 4 * sg, invK, T, gs, pl, s3);   // s3 = ground_seg3d_lines_world row (:569-578)
    float hitb[2][2];
    for (int e = 0; e < 2; e++) {
      float img[2][2];
      for (int c = 0; c < 2; c++) {                      // the ground point and the point 2 m above it (:419-427)
        const float Pw[3] = {s3[3 * e], s3[3 * e + 1], s3[3 * e + 2] + (c ? 2.f : 0.f)};
        float Ps[3], hh[3];
        for (int i = 0; i < 3; i++) Ps[i] = iT[i * 4 + 0] * Pw[0] + iT[i * 4 + 1] * Pw[1] + iT[i * 4 + 2] * Pw[2] + iT[i * 4 + 3] * 1.f;
        for (int i = 0; i < 3; i++) hh[i] = K[i * 3 + 0] * Ps[0] + K[i * 3 + 1] * Ps[1] + K[i * 3 + 2] * Ps[2];
        img[c][0] = hh[0] / hh[2]; img[c][1] = hh[1] / hh[2];
      }
      float dir[2] = {img[1][0] - img[0][0], img[1][1] - img[0][1]};
      if (dir[1] > 0) { dir[0] = -dir[0]; dir[1] = -dir[1]; }
      hit_boundary(seg2d + 4 * sg + 2 * e, dir, width, height, hitb[e]);
    }
    const float* p0 = seg2d + 4 * sg;
    const float* p1 = p0 + 2;
    const float* bh = hitb[0];
    const float* eh = hitb[1];
    float* v = verts + 2 * (size_t)nv;
    int k = 0;
    auto push = [&](float x, float y) { v[2 * k] = x; v[2 * k + 1] = y; k++; };
    push(p0[0], p0[1]); push(p1[0], p1[1]);
    if ((eh[0] != p1[0]) || (eh[1] != p1[1])) push(eh[0], eh[1]);
    if (0 < bh[0] && bh[0] < width - 1 && eh[0] == width - 1) push((float)(width - 1), 0.f);
    if (bh[0] == 0 && eh[0] == width - 1) { push((float)(width - 1), 0.f); push(0.f, 0.f); }
    if (bh[0] == 0 && 0 < eh[0] && eh[0] < width - 1) push(0.f, 0.f);
    if ((bh[0] != p0[0]) || (bh[1] != p0[1])) push(bh[0], bh[1]);
    push(p0[0], p0[1]);
    if ((bh[0] == -1) || (eh[0] == -1)) k = 0;
    nv += k;
    poly_off[sg + 2] = nv;
  }
  if (n_verts) *n_verts = nv;
  return PPS_OK;
}

// The interval derivation of k_popup_frame (pps_raster.h), compiled for the host and run row by row: what the kernel's
// setup phase computes, without a device.  Test hook only -- pps_popup_run never comes here.
int pps_popup_mask_host(const float* polys, const int* poly_off, int nplanes, int width, int height, int step, int32_t* plane_id) {
  if (!poly_off || !plane_id || nplanes < 0 || width <= 0 || height <= 0 || (step != 1 && step != 2)) return PPS_EINVAL;
  if (poly_off[nplanes] > 0 && !polys) return PPS_EINVAL;
  for (size_t i = 0; i < (size_t)width * height; i++) plane_id[i] = -1;
  const int Ws = (width + step - 1) / step;
  for (int p = 0; p < nplanes; p++) {
    const int v0 = poly_off[p], npts = poly_off[p + 1] - v0;
    if (npts < 1) continue;
    std::vector<float> P(2 * (size_t)npts);
    for (int i = 0; i < 2 * npts; i++) P[i] = step == 2 ? polys[2 * (size_t)v0 + i] / 2 : polys[2 * (size_t)v0 + i];
    int x0 = (int)P[0], x1 = x0, y0 = (int)P[1], y1 = y0;
    for (int i = 0; i < npts; i++) {
      const int x = (int)P[2 * i], y = (int)P[2 * i + 1];
      x0 = std::min(x0, x); x1 = std::max(x1, x); y0 = std::min(y0, y); y1 = std::max(y1, y);
    }
    const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
    std::vector<int2> q(npts);
    for (int i = 0; i < npts; i++) q[i] = make_int2((int)(P[2 * i] - (float)x0), (int)(P[2 * i + 1] - (float)y0));
    std::vector<RasterLine> lines(npts);
    for (int i = 0; i < npts; i++) { const int u = i > 0 ? i - 1 : npts - 1; lines[i] = raster_line(bw, bh, q[u].x, q[u].y, q[i].x, q[i].y); }
    for (int Y = 0; Y < height; Y += step) {
      const int cy = Y / step - y0;
      if (cy < 0 || cy >= bh) continue;
      for (int i = 0; i <= npts; i++) {
        int lo = 0, hi = -1;
        const bool hit = i == 0 ? raster_fill_row(q.data(), npts, bw, bh, cy, lo, hi) : raster_line_row(lines[i - 1], cy, lo, hi);
        if (!hit) continue;
        lo = std::max(std::max(lo, 0) + x0, 0); hi = std::min(std::min(hi, bw - 1) + x0, Ws - 1);
        for (int xs = lo; xs <= hi; xs++) plane_id[(size_t)Y * width + (size_t)xs * step] = p;
      }
    }
  }
  return PPS_OK;
}

int pps_popup_last_kernel_time(const pps_popup* p, double* sec) {
  if (!p || !sec) return PPS_EINVAL;
  *sec = p->last_kernel_s;
  return PPS_OK;
}

}  // extern "C"
