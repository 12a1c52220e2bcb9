// Ground-edge selection, device half + C-ABI (include/pps.h: pps_edges_*).
// Replaces popup_plane::edge_get_polygons (pop_up_wall/libs/select_edge.cpp:66-409) with its Python helpers
// (pop_up_python/python/pop_up_python/pop_up_fun.py:85-204).
//
// K7a k_label_close   [half-size nearest] -> dilate -> erode -> 255 - x in ONE pass over the label map
//                     (select_edge.cpp:69-78: cv::resize / dilate / erode / convertTo are four full-image passes
//                     there).  A 64 x 32 tile with the halo of both structuring elements is staged in LDS; the four
//                     separable max / min passes run LDS -> LDS, a thread producing four neighbouring window extrema
//                     from one shared partial result.  Algorithmic traffic: 1 B read + 1 B written per pixel.  The tile
//                     also classifies its marching-squares cells and adds their segment counts to a per-row counter.
// K7b k_cells_emit    the marching-squares cells of skimage.measure.find_contours(label, 0), in the raster order the
//                     Python code walks them (the contour linking that follows depends on that order): one workgroup
//                     per cell row starts at the prefix sum of the rows above it and orders the segments of its row
//                     with a workgroup scan.
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

constexpr int kTileX = 64, kTileY = 32, kCloseThreads = 256;
constexpr int kMaxElement = 31;   // largest structuring element side the LDS tile is sized for

struct CloseArgs {
  const unsigned char* src; int sw, sh;   // label map as given
  int half;                               // 1: work on the half-size nearest-neighbour copy (src(2x, 2y))
  unsigned char* dst; int w, h;           // pre-processed map
  int kd, ke;                             // dilate / erode element side
  int* row_count;                         // [h - 1] segments per cell row, zero on entry (atomically accumulated)
};

// LDS geometry of k_label_close for elements kd / ke: pitch and rows of one buffer (two buffers are used).  The output
// region is (kTileX + 1) x (kTileY + 1): one extra column / row so that the cells along the tile's right / lower
// border can be classified here; +3 because a thread produces strips of four values and may run past the end.
__host__ __device__ inline int close_pitch(int kd, int ke) { return (kTileX + 1 + (kd - 1) + (ke - 1) + 3 + 3) & ~3; }
__host__ __device__ inline int close_rows(int kd, int ke) { return kTileY + 1 + (kd - 1) + (ke - 1) + 3; }

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// four neighbouring window extrema at once: out[j] = op over p[(j + i) * stride], i < k.  The k - 3 values shared by
// all four windows are reduced once (k + 8 operations for four results instead of 4 k).
template <bool MAX, int K>
__device__ __forceinline__ void window4(const unsigned char* __restrict__ p, int stride, int k_rt, int out[4]) {
  const int k = K > 0 ? K : k_rt;
  auto op = [](int x, int y) { return MAX ? imax(x, y) : imin(x, y); };
  if (k >= 4) {
    int core = p[3 * stride];
    if (K > 0) {
#pragma unroll
      for (int i = 4; i < K; i++) core = op(core, p[i * stride]);
    } else {
      for (int i = 4; i < k; i++) core = op(core, p[i * stride]);
    }
    const int b0 = p[0], b1 = p[stride], b2 = p[2 * stride];
    const int c0 = p[k * stride], c1 = p[(k + 1) * stride], c2 = p[(k + 2) * stride];
    const int b12 = op(b1, b2), c01 = op(c0, c1);
    out[0] = op(core, op(b0, b12));
    out[1] = op(core, op(b12, c0));
    out[2] = op(core, op(b2, c01));
    out[3] = op(core, op(c01, c2));
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int v = p[j * stride];
      for (int i = 1; i < k; i++) v = op(v, p[(j + i) * stride]);
      out[j] = v;
    }
  }
}

__device__ __forceinline__ int cell_segments(const unsigned char* __restrict__ img, int w, int r0, int c0, pps_edges_host::CellSeg out[2]);

template <int KD, int KE>
__global__ __launch_bounds__(kCloseThreads) void k_label_close(CloseArgs a) {
  extern __shared__ unsigned char lds[];
  const int kd = KD > 0 ? KD : a.kd, ke = KE > 0 ? KE : a.ke;
  // element windows: [-ad, kd-1-ad] for the dilation, [-ae, ke-1-ae] for the erosion (default anchor = side / 2)
  const int ad = kd / 2, ae = ke / 2;
  const int OX = kTileX + 1, OY = kTileY + 1;              // closed values wanted (own pixels + the cell halo)
  const int DX = OX + (ke - 1), DY = OY + (ke - 1);        // dilated values wanted
  const int RX = DX + (kd - 1), RY = DY + (kd - 1);        // source region
  const int P = close_pitch(kd, ke);
  unsigned char* A = lds;
  unsigned char* B = lds + P * close_rows(kd, ke);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int x0 = blockIdx.x * kTileX, y0 = blockIdx.y * kTileY;
  const int sx0 = x0 - ae - ad, sy0 = y0 - ae - ad;
  // 1. source region -> A; outside the image 0, which the maximum ignores
#pragma unroll 4
  for (int yy = wv; yy < RY; yy += kCloseThreads / 64) {
    const int y = sy0 + yy;
    const bool yin = y >= 0 && y < a.h;
    const size_t row = (size_t)(a.half ? imin(2 * y, a.sh - 1) : y) * a.sw;
    for (int xx = lane; xx < RX; xx += 64) {
      const int x = sx0 + xx;
      unsigned char v = 0;
      if (yin && x >= 0 && x < a.w) v = a.src[row + (a.half ? imin(2 * x, a.sw - 1) : x)];
      A[yy * P + xx] = v;
    }
  }
  __syncthreads();
  int o[4];
  // 2. maximum along x: A -> B (RY rows x DX)
  {
    const int SX = (DX + 3) >> 2;
    for (int i = tid; i < RY * SX; i += kCloseThreads) {
      const int yy = i / SX, xs = (i - yy * SX) << 2;
      window4<true, KD>(A + yy * P + xs, 1, kd, o);
      *reinterpret_cast<uchar4*>(B + yy * P + xs) = make_uchar4((unsigned char)o[0], (unsigned char)o[1], (unsigned char)o[2], (unsigned char)o[3]);
    }
  }
  __syncthreads();
  // 3. maximum along y: B -> A (DY x DX); a position outside the image becomes 255, which the minimum ignores
  {
    const int SY = (DY + 3) >> 2;
    for (int i = tid; i < SY * DX; i += kCloseThreads) {
      const int ys = (i / DX) << 2, xx = i - (i / DX) * DX;
      window4<true, KD>(B + ys * P + xx, P, kd, o);
      const int x = x0 - ae + xx;
      const bool xin = x >= 0 && x < a.w;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int y = y0 - ae + ys + j;
        A[(ys + j) * P + xx] = (xin && y >= 0 && y < a.h) ? (unsigned char)o[j] : (unsigned char)255;
      }
    }
  }
  __syncthreads();
  // 4. minimum along x: A -> B (DY x OX)
  {
    const int SX = (OX + 3) >> 2;
    for (int i = tid; i < DY * SX; i += kCloseThreads) {
      const int yy = i / SX, xs = (i - yy * SX) << 2;
      window4<false, KE>(A + yy * P + xs, 1, ke, o);
      *reinterpret_cast<uchar4*>(B + yy * P + xs) = make_uchar4((unsigned char)o[0], (unsigned char)o[1], (unsigned char)o[2], (unsigned char)o[3]);
    }
  }
  __syncthreads();
  // 5. minimum along y, inverted (ground 255 -> 0): B -> A (OY x OX) and, for the tile's own pixels, -> dst
  {
    const int SY = (OY + 3) >> 2;
    for (int i = tid; i < SY * OX; i += kCloseThreads) {
      const int ys = (i / OX) << 2, xx = i - (i / OX) * OX;
      window4<false, KE>(B + ys * P + xx, P, ke, o);
      const int x = x0 + xx;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int yy = ys + j, y = y0 + yy;
        const unsigned char v = (unsigned char)(255 - o[j]);
        A[yy * P + xx] = v;
        if (xx < kTileX && yy < kTileY && x < a.w && y < a.h) a.dst[(size_t)y * a.w + x] = v;
      }
    }
  }
  __syncthreads();
  // 6. segments per cell row of this tile (k_cells_emit places the rows)
  for (int yy = wv; yy < kTileY; yy += kCloseThreads / 64) {
    const int y = y0 + yy, x = x0 + lane;
    if (y + 1 >= a.h) break;
    pps_edges_host::CellSeg tmp[2];
    int n = x + 1 < a.w ? cell_segments(A, P, yy, lane, tmp) : 0;
    for (int s = 32; s > 0; s >>= 1) n += __shfl_down(n, s);
    if (lane == 0 && n) atomicAdd(a.row_count + y, n);
  }
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

constexpr int kEmitRowsPerGroup = kRowThreads / 64;

// One wavefront per cell row, no LDS and no workgroup barrier: a lane owns a run of consecutive cells, counts their
// segments, a shuffle scan places the runs, and the lane classifies its cells a second time to write them in order.
__global__ __launch_bounds__(kRowThreads) void k_cells_emit(const unsigned char* __restrict__ img, int w, int h,
                                                            const int* __restrict__ row_count, int* __restrict__ next_row_count, int n_counters,
                                                            pps_edges_host::CellSeg* __restrict__ segs, pps_edges_host::CellSeg* __restrict__ host_segs,
                                                            int host_cap, int* __restrict__ host_total) {
  // the counters of the next call (k_label_close accumulates into zeros); all of them: the next map may be larger
  for (int r = blockIdx.x * kRowThreads + threadIdx.x; r < n_counters; r += gridDim.x * kRowThreads) next_row_count[r] = 0;
  const int lane = threadIdx.x & 63;
  // (round 6) the segments also go straight into pinned host memory -- the first host_cap of them, which is all of them for any real
  // contour -- and the first wave leaves their number there: the caller synchronises once and reads, no copy of the row counters and
  // no second copy sized by them
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    int tot = 0;
    for (int r = lane; r + 1 < h; r += 64) tot += row_count[r];
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0) host_total[0] = tot;
  }
  const int r0 = blockIdx.x * kEmitRowsPerGroup + (threadIdx.x >> 6);
  if (r0 + 1 >= h || row_count[r0] == 0) return;   // most rows hold no boundary
  int above = 0;
  for (int r = lane; r < r0; r += 64) above += row_count[r];
  for (int o = 32; o > 0; o >>= 1) above += __shfl_xor(above, o);
  const int per_lane = (w - 1 + 63) / 64;
  const int c_begin = lane * per_lane, c_end = imin(c_begin + per_lane, w - 1);
  pps_edges_host::CellSeg tmp[2];
  int n = 0;
  for (int c = c_begin; c < c_end; c++) n += cell_segments(img, w, r0, c, tmp);
  int incl = n;
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  int at = above + incl - n;
  if (n == 0) return;
  for (int c = c_begin; c < c_end; c++) {
    const int k = cell_segments(img, w, r0, c, tmp);
    if (k > 0) { segs[at] = tmp[0]; if (at < host_cap) host_segs[at] = tmp[0]; }
    if (k > 1) { segs[at + 1] = tmp[1]; if (at + 1 < host_cap) host_segs[at + 1] = tmp[1]; }
    at += k;
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
  int* d_row_count[2] = {nullptr, nullptr};   // height each; alternate between calls (the idle one is zeroed by k_cells_emit)
  int flip = 0;
  pps_edges_host::CellSeg* d_segs = nullptr;   // 2 (width - 1)(height - 1)
  pps_edges_host::CellSeg* h_segs = nullptr;   // pinned: the first h_cap segments as k_cells_emit writes them
  int h_cap = 0;
  int* h_total = nullptr;                      // pinned: number of segments of the last call
  int pre_w = 0, pre_h = 0;
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
  for (int k = 0; k < 2 && st == hipSuccess; k++) {
    st = hipMalloc(reinterpret_cast<void**>(&e->d_row_count[k]), sizeof(int) * (size_t)height);
    if (st == hipSuccess) st = hipMemset(e->d_row_count[k], 0, sizeof(int) * (size_t)height);
  }
  if (st == hipSuccess) st = hipMalloc(reinterpret_cast<void**>(&e->d_segs), sizeof(pps_edges_host::CellSeg) * 2 * px);
  e->h_cap = (int)std::min<size_t>(2 * px, (size_t)1 << 18);        // 2 MB of pinned memory at most; more segments than that take the copy
  if (st == hipSuccess) st = hipHostMalloc(reinterpret_cast<void**>(&e->h_segs), sizeof(pps_edges_host::CellSeg) * (size_t)e->h_cap, hipHostMallocDefault);
  if (st == hipSuccess) st = hipHostMalloc(reinterpret_cast<void**>(&e->h_total), 2 * sizeof(int), hipHostMallocDefault);
  if (st != hipSuccess) { pps_edges_destroy(e); return PPS_EHIP; }
  *out = e;
  return PPS_OK;
}

int pps_edges_destroy(pps_edges* e) {
  if (!e) return PPS_OK;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  (void)hipFree(e->d_label); (void)hipFree(e->d_pre); (void)hipFree(e->d_row_count[0]); (void)hipFree(e->d_row_count[1]); (void)hipFree(e->d_segs);
  if (e->h_segs) (void)hipHostFree(e->h_segs);
  if (e->h_total) (void)hipHostFree(e->h_total);
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
  int* rc = e->d_row_count[e->flip];
  int* rc_next = e->d_row_count[e->flip ^ 1];
  e->flip ^= 1;
  CloseArgs ca{d_src, e->width, e->height, prm.downsample_contour ? 1 : 0, e->d_pre, w, h, kd, ke, rc};
  const dim3 grid((w + kTileX - 1) / kTileX, (h + kTileY - 1) / kTileY);
  const size_t lds_bytes = 2 * (size_t)close_pitch(kd, ke) * close_rows(kd, ke);
  EHIP(e, hipEventRecord(e->ev[0], e->stream));
  if (kd == 11 && ke == 11) hipLaunchKernelGGL((k_label_close<11, 11>), grid, dim3(kCloseThreads), lds_bytes, e->stream, ca);
  else if (kd == 8 && ke == 8) hipLaunchKernelGGL((k_label_close<8, 8>), grid, dim3(kCloseThreads), lds_bytes, e->stream, ca);
  else hipLaunchKernelGGL((k_label_close<0, 0>), grid, dim3(kCloseThreads), lds_bytes, e->stream, ca);
  hipLaunchKernelGGL(k_cells_emit, dim3((h - 1 + kEmitRowsPerGroup - 1) / kEmitRowsPerGroup), dim3(kRowThreads), 0, e->stream, e->d_pre, w, h, rc, rc_next, e->height, e->d_segs, e->h_segs, e->h_cap, e->h_total);
  EHIP(e, hipGetLastError());
  EHIP(e, hipEventRecord(e->ev[1], e->stream));
  try {
    EHIP(e, hipStreamSynchronize(e->stream));                    // the one synchronisation of the call: count and segments are in pinned memory
    const size_t nseg = (size_t)std::max(0, e->h_total[0]);
    const pps_edges_host::CellSeg* segs = e->h_segs;
    if (nseg > (size_t)e->h_cap) {                                // (more boundary cells than the pinned block holds: the copy, sized by the count)
      e->segs.resize(nseg);
      EHIP(e, hipMemcpy(e->segs.data(), e->d_segs, sizeof(pps_edges_host::CellSeg) * nseg, hipMemcpyDeviceToHost));
      segs = e->segs.data();
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, e->ev[0], e->ev[1]) == hipSuccess) e->last_kernel_s = 1e-3 * ms;
    e->contour = pps_edges_host::ground_contour(segs, (int)nseg, prm.downsample_contour ? 2.0f : 1.0f);
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
