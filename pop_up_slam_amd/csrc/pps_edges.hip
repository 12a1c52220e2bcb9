// Ground-edge selection, device half + C-ABI (include/pps.h: pps_edges_*).
// Replaces popup_plane::edge_get_polygons (pop_up_wall/libs/select_edge.cpp:66-409) with its Python helpers
// (pop_up_python/python/pop_up_python/pop_up_fun.py:85-204).
//
// K7a k_label_close   [half-size nearest] -> dilate -> erode -> 255 - x in ONE pass over the label map
//                     (select_edge.cpp:69-78: cv::resize / dilate / erode / convertTo are four full-image passes
//                     there).  A 64 x 16 tile with the halo of both structuring elements is staged in LDS; the four
//                     separable max / min passes run LDS -> LDS.  Algorithmic traffic: 1 B read + 1 B written per pixel.
// K7b k_cells_count / k_cells_emit   the marching-squares cells of skimage.measure.find_contours(label, 0), in the
//                     raster order the Python code walks them (the contour linking that follows depends on that
//                     order): one workgroup per cell row counts its segments, a second launch places every row at the
//                     prefix sum of the rows above it and orders the segments of a row with a workgroup scan.
// The linking of the segments into contours and the selection itself are host work (pps_edges_host.cpp).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/pps.h"
#include "pps_edges_host.h"

namespace {

constexpr int kTileX = 64, kTileY = 16, kCloseThreads = 256;
constexpr int kMaxElement = 31;   // largest structuring element side the LDS tile is sized for

struct CloseArgs {
  const unsigned char* src; int sw, sh;   // label map as given
  int half;                               // 1: work on the half-size nearest-neighbour copy (src(2x, 2y))
  unsigned char* dst; int w, h;           // pre-processed map
  int kd, ke;                             // dilate / erode element side
};

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

__global__ __launch_bounds__(kCloseThreads) void k_label_close(CloseArgs a) {
  extern __shared__ unsigned char lds[];
  // element windows: [-ad, bd] for the dilation, [-ae, be] for the erosion (default anchor = side / 2)
  const int ad = a.kd / 2, bd = a.kd - 1 - ad, ae = a.ke / 2, be = a.ke - 1 - ae;
  const int RX = kTileX + (a.kd - 1) + (a.ke - 1), RY = kTileY + (a.kd - 1) + (a.ke - 1);   // source region
  const int DX = kTileX + (a.ke - 1), DY = kTileY + (a.ke - 1);                               // dilated region
  unsigned char* A = lds;
  unsigned char* B = lds + RX * RY;
  const int x0 = blockIdx.x * kTileX, y0 = blockIdx.y * kTileY;
  const int sx0 = x0 - ae - ad, sy0 = y0 - ae - ad;
  // 1. source region; outside the image 0, which the maximum ignores
  for (int i = threadIdx.x; i < RX * RY; i += kCloseThreads) {
    const int yy = i / RX, xx = i - yy * RX;
    const int x = sx0 + xx, y = sy0 + yy;
    unsigned char v = 0;
    if (x >= 0 && x < a.w && y >= 0 && y < a.h) {
      const int sx = a.half ? imin(2 * x, a.sw - 1) : x, sy = a.half ? imin(2 * y, a.sh - 1) : y;
      v = a.src[(size_t)sy * a.sw + sx];
    }
    A[i] = v;
  }
  __syncthreads();
  // 2. maximum along x: B (RY x DX)
  for (int i = threadIdx.x; i < RY * DX; i += kCloseThreads) {
    const int yy = i / DX, xx = i - yy * DX;
    int v = 0;
    for (int j = 0; j < a.kd; j++) v = imax(v, A[yy * RX + xx + j]);
    B[i] = (unsigned char)v;
  }
  __syncthreads();
  // 3. maximum along y: A (DY x DX); a position outside the image is 255, which the minimum ignores
  for (int i = threadIdx.x; i < DY * DX; i += kCloseThreads) {
    const int yy = i / DX, xx = i - yy * DX;
    const int x = x0 - ae + xx, y = y0 - ae + yy;
    int v = 0;
    for (int j = 0; j < a.kd; j++) v = imax(v, B[(yy + j) * DX + xx]);
    A[i] = (x >= 0 && x < a.w && y >= 0 && y < a.h) ? (unsigned char)v : (unsigned char)255;
  }
  __syncthreads();
  // 4. minimum along x: B (DY x kTileX)
  for (int i = threadIdx.x; i < DY * kTileX; i += kCloseThreads) {
    const int yy = i / kTileX, xx = i - yy * kTileX;
    int v = 255;
    for (int j = 0; j < a.ke; j++) v = imin(v, A[yy * DX + xx + j]);
    B[i] = (unsigned char)v;
  }
  __syncthreads();
  // 5. minimum along y, inverted: ground 255 -> 0
  for (int i = threadIdx.x; i < kTileY * kTileX; i += kCloseThreads) {
    const int yy = i / kTileX, xx = i - yy * kTileX;
    const int x = x0 + xx, y = y0 + yy;
    if (x >= a.w || y >= a.h) continue;
    int v = 255;
    for (int j = 0; j < a.ke; j++) v = imin(v, B[(yy + j) * kTileX + xx]);
    a.dst[(size_t)y * a.w + x] = (unsigned char)(255 - v);
  }
  (void)bd; (void)be;
}

// The segments of one 2x2 cell at level 0: a vertex is "high" when > 0 and an edge crossing sits on the zero end of
// its edge, so every end point is a pixel centre.  Returns the number of non-degenerate segments (0..2).
__device__ __forceinline__ int cell_segments(const unsigned char* __restrict__ img, int w, int r0, int c0,
                                             pps_edges_host::CellSeg out[2]) {
  const int r1 = r0 + 1, c1 = c0 + 1;
  const bool ul = img[(size_t)r0 * w + c0] > 0, ur = img[(size_t)r0 * w + c1] > 0, ll = img[(size_t)r1 * w + c0] > 0,
             lr = img[(size_t)r1 * w + c1] > 0;
  const int sq = (ul ? 1 : 0) | (ur ? 2 : 0) | (ll ? 4 : 0) | (lr ? 8 : 0);
  if (sq == 0 || sq == 15) return 0;
  // crossing points of the four edges as (row, column): T top, B bottom, L left, R right
  const short Tr = (short)r0, Tc = (short)(ul ? c1 : c0);
  const short Br = (short)r1, Bc = (short)(ll ? c1 : c0);
  const short Lr = (short)(ul ? r1 : r0), Lc = (short)c0;
  const short Rr = (short)(ur ? r1 : r0), Rc = (short)c1;
  // from -> to per case ('low' vertex connectivity on the two saddle cases)
  //                 0  1  2  3  4  5  6  7  8  9 10 11 12 13 14
  // edge codes: 0 = T, 1 = B, 2 = L, 3 = R
  constexpr unsigned char F0[16] = {0, 0, 3, 3, 2, 0, 3, 3, 1, 0, 1, 1, 2, 0, 2, 0};
  constexpr unsigned char T0[16] = {0, 2, 0, 2, 1, 1, 0, 1, 3, 2, 0, 2, 3, 3, 0, 0};
  const short er[4] = {Tr, Br, Lr, Rr}, ec[4] = {Tc, Bc, Lc, Rc};
  int n = 0;
  {
    const int f = F0[sq], t = T0[sq];
    if (er[f] != er[t] || ec[f] != ec[t]) { out[n].fr = er[f]; out[n].fc = ec[f]; out[n].tr = er[t]; out[n].tc = ec[t]; n++; }
  }
  if (sq == 6) {        // second arc: left -> bottom
    if (Lr != Br || Lc != Bc) { out[n].fr = Lr; out[n].fc = Lc; out[n].tr = Br; out[n].tc = Bc; n++; }
  } else if (sq == 9) { // second arc: bottom -> right
    if (Br != Rr || Bc != Rc) { out[n].fr = Br; out[n].fc = Bc; out[n].tr = Rr; out[n].tc = Rc; n++; }
  }
  return n;
}

constexpr int kRowThreads = 256;

__device__ __forceinline__ int block_sum(int v, int* scratch) {   // all threads get the total
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  int t = 0;
  for (int k = 0; k < kRowThreads / 64; k++) t += scratch[k];
  return t;
}

__global__ __launch_bounds__(kRowThreads) void k_cells_count(const unsigned char* __restrict__ img, int w, int h, int* __restrict__ row_count) {
  __shared__ int scratch[kRowThreads / 64];
  const int r0 = blockIdx.x;
  int n = 0;
  pps_edges_host::CellSeg tmp[2];
  for (int c0 = threadIdx.x; c0 + 1 < w; c0 += kRowThreads) n += cell_segments(img, w, r0, c0, tmp);
  const int total = block_sum(n, scratch);
  if (threadIdx.x == 0) row_count[r0] = total;
}

__global__ __launch_bounds__(kRowThreads) void k_cells_emit(const unsigned char* __restrict__ img, int w, int h,
                                                            const int* __restrict__ row_count,
                                                            pps_edges_host::CellSeg* __restrict__ segs) {
  __shared__ int scratch[kRowThreads / 64];
  __shared__ int wave_off[kRowThreads / 64];
  const int r0 = blockIdx.x;
  int above = 0;
  for (int r = threadIdx.x; r < r0; r += kRowThreads) above += row_count[r];
  int base = block_sum(above, scratch);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int cb = 0; cb + 1 < w; cb += kRowThreads) {
    const int c0 = cb + threadIdx.x;
    pps_edges_host::CellSeg mine[2];
    const int n = c0 + 1 < w ? cell_segments(img, w, r0, c0, mine) : 0;
    // exclusive scan over the workgroup: inside a wave by shuffles, across waves through LDS
    int incl = n;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    __syncthreads();
    if (lane == 63) wave_off[wave] = incl;
    __syncthreads();
    int before = 0, chunk = 0;
    for (int k = 0; k < kRowThreads / 64; k++) { if (k < wave) before += wave_off[k]; chunk += wave_off[k]; }
    const int at = base + before + incl - n;
    for (int k = 0; k < n; k++) segs[at + k] = mine[k];
    base += chunk;
  }
}

}  // namespace

struct pps_edges {
  int device = 0, width = 0, height = 0;
  std::string err;
  hipStream_t stream = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  unsigned char* d_label = nullptr;   // width x height
  unsigned char* d_pre = nullptr;     // pre-processed map (<= width x height)
  int* d_row_count = nullptr;         // height - 1
  pps_edges_host::CellSeg* d_segs = nullptr;   // 2 (width - 1)(height - 1)
  int pre_w = 0, pre_h = 0;
  std::vector<int> row_count;
  std::vector<pps_edges_host::CellSeg> segs;
  pps_edges_host::Contour contour;
  double last_kernel_s = 0;
};

namespace {
int efail(pps_edges* e, int code, const std::string& m) { if (e) e->err = m; return code; }
#define EHIP(e, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return efail(e, PPS_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)
}  // namespace

extern "C" {

void pps_edge_default_params(pps_edge_params* p) {
  if (!p) return;
  p->downsample_contour = 0;
  p->dilation_distance = 11; p->erosion_distance = 11;
  p->pre_vertical_thre = 15; p->pre_minium_len = 15; p->pre_contour_close_thre = 50; p->interval_overlap_thre = 20;
  p->post_short_thre = 30; p->post_bind_dist_thre = 10; p->post_merge_dist_thre = 20; p->post_merge_angle_thre = 10;
  p->post_extend_thre = 15;
  p->pre_boundary_thre = 5; p->pre_merge_angle_thre = 10; p->pre_merge_dist_thre = 10; p->pre_proj_angle_thre = 20;
  p->pre_proj_cover_thre = 0.6; p->pre_proj_cover_large_thre = 0.8; p->pre_proj_dist_thre = 100;
}

int pps_edges_create(int device, int width, int height, pps_edges** out) {
  if (!out || width < 2 || height < 2 || width > 32767 || height > 32767) return PPS_EINVAL;
  *out = nullptr;
  pps_edges* e = new (std::nothrow) pps_edges;
  if (!e) return PPS_ENOMEM;
  e->device = device; e->width = width; e->height = height;
  const size_t px = (size_t)width * height;
  hipError_t st = hipSetDevice(device);
  if (st == hipSuccess) st = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (st == hipSuccess) st = hipEventCreate(&e->ev[0]);
  if (st == hipSuccess) st = hipEventCreate(&e->ev[1]);
  if (st == hipSuccess) st = hipMalloc(reinterpret_cast<void**>(&e->d_label), px);
  if (st == hipSuccess) st = hipMalloc(reinterpret_cast<void**>(&e->d_pre), px);
  if (st == hipSuccess) st = hipMalloc(reinterpret_cast<void**>(&e->d_row_count), sizeof(int) * (size_t)height);
  if (st == hipSuccess) st = hipMalloc(reinterpret_cast<void**>(&e->d_segs), sizeof(pps_edges_host::CellSeg) * 2 * px);
  if (st != hipSuccess) { pps_edges_destroy(e); return PPS_EHIP; }
  *out = e;
  return PPS_OK;
}

int pps_edges_destroy(pps_edges* e) {
  if (!e) return PPS_OK;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  (void)hipFree(e->d_label); (void)hipFree(e->d_pre); (void)hipFree(e->d_row_count); (void)hipFree(e->d_segs);
  if (e->ev[0]) (void)hipEventDestroy(e->ev[0]);
  if (e->ev[1]) (void)hipEventDestroy(e->ev[1]);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return PPS_OK;
}

const char* pps_edges_last_error(const pps_edges* e) { return e ? e->err.c_str() : "null handle"; }

int pps_edges_select(pps_edges* e, const unsigned char* label_map, int label_on_device, const float* lsd_lines, int n_lines,
                     const pps_edge_params* prm_in, float* open_segs, int* n_open, float* closed_segs, int* n_closed,
                     float* open_in_closed) {
  if (!e) return PPS_EINVAL;
  if (!label_map || n_lines < 0 || (n_lines > 0 && !lsd_lines) || !open_segs || !n_open || !closed_segs || !n_closed || !open_in_closed)
    return efail(e, PPS_EINVAL, "pps_edges_select: null argument");
  pps_edge_params prm;
  if (prm_in) prm = *prm_in; else pps_edge_default_params(&prm);
  const int kd = prm.downsample_contour ? 8 : prm.dilation_distance, ke = prm.downsample_contour ? 8 : prm.erosion_distance;
  if (kd < 1 || ke < 1 || kd > kMaxElement || ke > kMaxElement) return efail(e, PPS_EINVAL, "structuring element side out of range (1..31)");
  *n_open = 0; *n_closed = 0;
  EHIP(e, hipSetDevice(e->device));
  const unsigned char* d_src = label_map;
  if (!label_on_device) {
    EHIP(e, hipMemcpyAsync(e->d_label, label_map, (size_t)e->width * e->height, hipMemcpyHostToDevice, e->stream));
    d_src = e->d_label;
  }
  // cv::resize(..., 0.5, 0.5): dsize = round(size * 0.5)
  const int w = prm.downsample_contour ? (int)std::lrint(e->width * 0.5) : e->width;
  const int h = prm.downsample_contour ? (int)std::lrint(e->height * 0.5) : e->height;
  if (w < 2 || h < 2) return efail(e, PPS_EINVAL, "label map too small");
  e->pre_w = w; e->pre_h = h;
  CloseArgs ca{d_src, e->width, e->height, prm.downsample_contour ? 1 : 0, e->d_pre, w, h, kd, ke};
  const int RX = kTileX + kd - 1 + ke - 1, RY = kTileY + kd - 1 + ke - 1;
  EHIP(e, hipEventRecord(e->ev[0], e->stream));
  hipLaunchKernelGGL(k_label_close, dim3((w + kTileX - 1) / kTileX, (h + kTileY - 1) / kTileY), dim3(kCloseThreads), 2 * (size_t)RX * RY,
                     e->stream, ca);
  hipLaunchKernelGGL(k_cells_count, dim3(h - 1), dim3(kRowThreads), 0, e->stream, e->d_pre, w, h, e->d_row_count);
  hipLaunchKernelGGL(k_cells_emit, dim3(h - 1), dim3(kRowThreads), 0, e->stream, e->d_pre, w, h, e->d_row_count, e->d_segs);
  EHIP(e, hipGetLastError());
  EHIP(e, hipEventRecord(e->ev[1], e->stream));
  try {
    e->row_count.resize((size_t)h - 1);
    EHIP(e, hipMemcpyAsync(e->row_count.data(), e->d_row_count, sizeof(int) * (size_t)(h - 1), hipMemcpyDeviceToHost, e->stream));
    EHIP(e, hipStreamSynchronize(e->stream));
    size_t nseg = 0;
    for (int c : e->row_count) nseg += (size_t)c;
    e->segs.resize(nseg);
    if (nseg) {
      EHIP(e, hipMemcpyAsync(e->segs.data(), e->d_segs, sizeof(pps_edges_host::CellSeg) * nseg, hipMemcpyDeviceToHost, e->stream));
      EHIP(e, hipStreamSynchronize(e->stream));
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, e->ev[0], e->ev[1]) == hipSuccess) e->last_kernel_s = 1e-3 * ms;
    e->contour = pps_edges_host::ground_contour(e->segs.data(), (int)nseg, prm.downsample_contour ? 2.0f : 1.0f);
    if (e->contour.xy.empty()) return PPS_OK;
    const pps_edges_host::Selection sel = pps_edges_host::select(e->contour.xy, e->width, e->height, lsd_lines, n_lines, prm);
    *n_open = (int)sel.open_segs.size() / 4;
    *n_closed = (int)sel.closed_segs.size() / 4;
    if (*n_open) std::memcpy(open_segs, sel.open_segs.data(), sizeof(float) * sel.open_segs.size());
    if (*n_closed) std::memcpy(closed_segs, sel.closed_segs.data(), sizeof(float) * sel.closed_segs.size());
    if (*n_open) std::memcpy(open_in_closed, sel.open_in_closed.data(), sizeof(float) * sel.open_in_closed.size());
  } catch (const std::bad_alloc&) {
    return efail(e, PPS_ENOMEM, "out of host memory");
  }
  return PPS_OK;
}

int pps_edges_download_label(pps_edges* e, unsigned char* out, int* w, int* h) {
  if (!e || !out) return PPS_EINVAL;
  if (e->pre_w == 0) return efail(e, PPS_ESTATE, "no pps_edges_select yet");
  EHIP(e, hipSetDevice(e->device));
  EHIP(e, hipMemcpy(out, e->d_pre, (size_t)e->pre_w * e->pre_h, hipMemcpyDeviceToHost));
  if (w) *w = e->pre_w;
  if (h) *h = e->pre_h;
  return PPS_OK;
}

int pps_edges_contour(pps_edges* e, float* xy, int cap, int* n, int* n_contours, int* n_points) {
  if (!e || !n || cap < 0 || (cap > 0 && !xy)) return PPS_EINVAL;
  const int have = (int)e->contour.xy.size() / 2;
  *n = have;
  if (n_contours) *n_contours = e->contour.n_contours;
  if (n_points) *n_points = e->contour.n_points;
  const int m = have < cap ? have : cap;
  if (m) std::memcpy(xy, e->contour.xy.data(), sizeof(float) * 2 * (size_t)m);
  return PPS_OK;
}

int pps_edges_last_kernel_time(const pps_edges* e, double* sec) {
  if (!e || !sec) return PPS_EINVAL;
  *sec = e->last_kernel_s;
  return PPS_OK;
}

int pps_edges_host_contour(const int16_t* cell_segs, int n, float scale, float* xy, int cap, int* n_xy, int* n_contours, int* n_points) {
  if (n < 0 || (n > 0 && !cell_segs) || !n_xy || cap < 0 || (cap > 0 && !xy)) return PPS_EINVAL;
  static_assert(sizeof(pps_edges_host::CellSeg) == 4 * sizeof(int16_t), "CellSeg is four int16");
  try {
    const pps_edges_host::Contour c = pps_edges_host::ground_contour(reinterpret_cast<const pps_edges_host::CellSeg*>(cell_segs), n, scale);
    *n_xy = (int)c.xy.size() / 2;
    if (n_contours) *n_contours = c.n_contours;
    if (n_points) *n_points = c.n_points;
    const int m = *n_xy < cap ? *n_xy : cap;
    if (m) std::memcpy(xy, c.xy.data(), sizeof(float) * 2 * (size_t)m);
  } catch (const std::bad_alloc&) { return PPS_ENOMEM; }
  return PPS_OK;
}

int pps_edges_host_select(const float* contour_xy, int n_contour, int width, int height, const float* lsd_lines, int n_lines,
                          const pps_edge_params* prm_in, float* open_segs, int* n_open, float* closed_segs, int* n_closed,
                          float* open_in_closed) {
  if (n_contour < 0 || (n_contour > 0 && !contour_xy) || n_lines < 0 || (n_lines > 0 && !lsd_lines) || !open_segs || !n_open ||
      !closed_segs || !n_closed || !open_in_closed)
    return PPS_EINVAL;
  pps_edge_params prm;
  if (prm_in) prm = *prm_in; else pps_edge_default_params(&prm);
  *n_open = 0; *n_closed = 0;
  if (n_contour == 0) return PPS_OK;
  try {
    const std::vector<float> cxy(contour_xy, contour_xy + 2 * (size_t)n_contour);
    const pps_edges_host::Selection sel = pps_edges_host::select(cxy, width, height, lsd_lines, n_lines, prm);
    *n_open = (int)sel.open_segs.size() / 4;
    *n_closed = (int)sel.closed_segs.size() / 4;
    if (*n_open) std::memcpy(open_segs, sel.open_segs.data(), sizeof(float) * sel.open_segs.size());
    if (*n_closed) std::memcpy(closed_segs, sel.closed_segs.data(), sizeof(float) * sel.closed_segs.size());
    if (*n_open) std::memcpy(open_in_closed, sel.open_in_closed.data(), sizeof(float) * sel.open_in_closed.size());
  } catch (const std::bad_alloc&) { return PPS_ENOMEM; }
  return PPS_OK;
}

}  // extern "C"
