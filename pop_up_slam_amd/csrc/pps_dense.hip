// pps_dense.hip -- K3 for large fronts ("dense-front" form of the multifrontal Cholesky).
//
// Pose graphs with loop closures (a loop-closing edge crosses every cut between its end points) and
// `loopclose_merge` (Mapping.cpp:636-709) produce separators of hundreds of scalars: fronts of several
// hundred rows that neither fit a wavefront's registers nor LDS.  Such a front lives where its results
// live -- the factor panel L ((f+1) x p, row-major, last row = right-hand side) and the update matrix U
// ((b+1) x (b+1), row-major lower triangle) -- and every step is spread over many workgroups:
//
//   once per factorisation   memset(L)  +  k_dense_hpush     original entries (H in front gather order) -> L panels
//   per tree level           k_dense_assemble   pull the children's update matrices into L (+=) and U (=)
//                            k_dense_panel      L_A = chol(A) (every workgroup redundantly, in LDS), then one
//                                               thread per row:  L_B = B L_A^-T  (rhs row included)
//                            k_dense_trailing   U -= L_B L_B^T, 64x64 tiles, v_mfma_f64_16x16x4_f64
//   per tree level (reverse) k_dense_solve      x_p = L_A^-T (y - L_B^T x_b)
//
// Same arithmetic as CholeskyImpl::factorize (Thirdparty/isam/isamlib/Cholesky.cpp:68-147): damped normal
// equations, L L^T, forward and backward solves; the summation order differs from CHOLMOD's (as every
// supernodal ordering does), the result agrees to round-off (tests/test_gpu_datasets.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "pps_device.h"
#include "pps_regtile.h"

namespace pps {
namespace {

constexpr int kMaxPiv = 64;            // pivots per front (AnalysisParams::max_pivots <= 64)
constexpr int kLdA = kMaxPiv + 1;


__device__ __forceinline__ int tri_row(int t) {           // largest r with r(r+1)/2 <= t
  int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((r + 1) * (r + 2) / 2 <= t) r++;
  while (r * (r + 1) / 2 > t) r--;
  return r;
}

// Workgroup -> (front of the level, work item inside the front): `off` is the level's prefix sum of per-front work
// item counts (count+1 entries, wave-uniform binary search), so that no empty workgroups are launched.
__device__ __forceinline__ int locate(const int* __restrict__ off, int count, int wg, int* item) {
  int lo = 0, hi = count;                 // invariant: off[lo] <= wg < off[hi]
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= wg) lo = mid; else hi = mid; }
  *item = wg - off[lo];
  return lo;
}

// ---- original entries of every front -> its (zeroed) factor panel --------------------------------------------------
// Hf[e] belongs at packed-triangle index el_tgt[e] of the front; all such entries sit in pivot columns (an H block
// is assembled where its column node is eliminated) or in the rhs row.  Diagonal entries carry bit 30: LM damping
// multiplies them by (1 + lambda) (Cholesky.cpp:94-97).
__global__ __launch_bounds__(256) void k_dense_hpush(DevGraph d, double lambda) {
  const int s = blockIdx.y;
  const int e0 = d.f_el_off[s], e1 = d.f_el_off[s + 1];
  const int p = d.f_p[s];
  double* __restrict__ Lp = d.L + d.f_Loff[s];
  const double damp = 1.0 + lambda;
  for (int e = e0 + blockIdx.x * 256 + threadIdx.x; e < e1; e += gridDim.x * 256) {
    const int tg = d.el_tgt[e];
    const int t = tg & 0x3fffffff;
    const int r = tri_row(t), c = t - r * (r + 1) / 2;
    const double v = d.Hf[e];
    Lp[(size_t)r * p + c] += (tg & (1 << 30)) ? v * damp : v;      // targets are unique inside a front
  }
}

// ---- extend-add into the factor panel, pull form: one 32x32 tile of the (f+1) x p panel per workgroup ------------------
// (the update-matrix part of the front is pulled by k_dense_trailing, which then writes U exactly once)
__global__ __launch_bounds__(256) void k_dense_assemble(DevGraph d, int level_begin, const int* __restrict__ off, int count) {
  __shared__ int invr[32], invc[32];
  int item;
  const int s = d.level_fronts[level_begin + locate(off, count, blockIdx.x, &item)];
  const int p = d.f_p[s], b = d.f_b[s], fa = p + b + 1;
  const int TC = (p + 31) / 32;
  const int ti = item / TC, tj = item - ti * TC;
  if (tj > ti) return;                               // entirely above the diagonal
  const int r0 = ti * 32, c0 = tj * 32;
  const int tid = threadIdx.x;
  const int cc = c0 + (tid & 31);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int ci = d.f_child_off[s]; ci < d.f_child_off[s + 1]; ci++) {
    const int c = d.child[ci];
    const int bc1 = d.f_b[c] + 1;
    const int* __restrict__ cm = d.cmap + d.f_cmap_off[c];
    const double* __restrict__ Uc = d.U + d.f_Uoff[c];
    if (tid < 32) { invr[tid] = -1; invc[tid] = -1; }
    __syncthreads();
    for (int i = tid; i < bc1; i += 256) {
      const int m = cm[i];
      if (m >= r0 && m < r0 + 32) invr[m - r0] = i;
      if (m >= c0 && m < c0 + 32) invc[m - c0] = i;
    }
    __syncthreads();
    const int ic = invc[tid & 31];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rl = (tid >> 5) + 8 * e;
      const int ir = invr[rl];
      if (ir >= 0 && ic >= 0 && r0 + rl >= cc) {
        const int hi = ir > ic ? ir : ic, lo = ir > ic ? ic : ir;
        acc[e] += Uc[(size_t)hi * bc1 + lo];
      }
    }
    __syncthreads();
  }
  double* __restrict__ Lp = d.L + d.f_Loff[s];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int rr = r0 + (tid >> 5) + 8 * e;
    if (rr < fa && cc <= rr && cc < p && acc[e] != 0.0) Lp[(size_t)rr * p + cc] += acc[e];
  }
}

// ---- panel: L_A = chol(A) (all workgroups of a front compute the same L_A), then L_B = B L_A^-T --------------------------
// The p x p diagonal block (p <= 64) is factored by wave 0 alone, register-resident: ten 16x16 tiles in MFMA
// accumulator layout, 4-column panels through a 64 x 5 LDS buffer, 4x4 diagonal blocks broadcast with v_readlane,
// rank-4 trailing updates as single v_mfma_f64_16x16x4_f64 (pps_regtile.h) -- no workgroup barrier inside.
__global__ __launch_bounds__(256) void k_dense_panel(DevGraph d, int level_begin, const int* __restrict__ off, int count) {
  __shared__ double A[kMaxPiv * kLdA];
  __shared__ double rinv[kMaxPiv];
  __shared__ double P[64 * kPStride];
  int slab;
  const int s = d.level_fronts[level_begin + locate(off, count, blockIdx.x, &slab)];
  const int p = d.f_p[s], b = d.f_b[s], fa = p + b + 1;
  double* __restrict__ Lp = d.L + d.f_Loff[s];
  const int tid = threadIdx.x;
  for (int i = tid; i < kMaxPiv * kLdA; i += 256) A[i] = 0.0;
  __syncthreads();
  if (tid < 64) {
    const int lane = tid, l16 = lane & 15, lq = lane >> 4;
    double4_t c[10];
#pragma unroll
    for (int ti = 0; ti < 4; ti++)
#pragma unroll
      for (int tj = 0; tj <= ti; tj++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 16 * ti + lq + 4 * r, col = 16 * tj + l16;
          const bool ok = row < p && col <= row;
          const double x = Lp[ok ? (size_t)row * p + col : 0];
          c[tile_id(ti, tj)][r] = ok ? x : 0.0;
        }
    for (int K = 0; K < p; K += 4) {
      const int nb = p - K < 4 ? p - K : 4;
      const int tjK = K >> 4, c0 = K & 15;
      switch (tjK) {
        case 0: reg_extract_panel<0>(c, P, c0, lane); break;
        case 1: reg_extract_panel<1>(c, P, c0, lane); break;
        case 2: reg_extract_panel<2>(c, P, c0, lane); break;
        default: reg_extract_panel<3>(c, P, c0, lane); break;
      }
      __builtin_amdgcn_wave_barrier();
      double r0 = P[lane * kPStride + 0], r1 = P[lane * kPStride + 1], r2 = P[lane * kPStride + 2], r3 = P[lane * kPStride + 3];
      const double d00 = readlane_d(r0, K);
      const double d10 = readlane_d(r0, K + 1), d11 = readlane_d(r1, K + 1);
      const double d20 = readlane_d(r0, K + 2), d21 = readlane_d(r1, K + 2), d22 = readlane_d(r2, K + 2);
      const double d30 = readlane_d(r0, K + 3), d31 = readlane_d(r1, K + 3), d32 = readlane_d(r2, K + 3), d33 = readlane_d(r3, K + 3);
      bool bad = false;
      double i0 = 0, i1 = 0, i2 = 0, i3 = 0, l10 = 0, l20 = 0, l30 = 0, l21 = 0, l31 = 0, l32 = 0;
      { bad |= !(d00 > 0.0); i0 = d00 > 0.0 ? rsqrt_nr(d00) : 0.0; l10 = d10 * i0; l20 = d20 * i0; l30 = d30 * i0; }
      if (nb > 1) { const double t = d11 - l10 * l10; bad |= !(t > 0.0); i1 = t > 0.0 ? rsqrt_nr(t) : 0.0; l21 = (d21 - l20 * l10) * i1; l31 = (d31 - l30 * l10) * i1; }
      if (nb > 2) { const double t = d22 - l20 * l20 - l21 * l21; bad |= !(t > 0.0); i2 = t > 0.0 ? rsqrt_nr(t) : 0.0; l32 = (d32 - l30 * l20 - l31 * l21) * i2; }
      if (nb > 3) { const double t = d33 - l30 * l30 - l31 * l31 - l32 * l32; bad |= !(t > 0.0); i3 = t > 0.0 ? rsqrt_nr(t) : 0.0; }
      if (bad && lane == 0 && slab == 0) raise_status(&d.result_dev[2], 1.0);                // not positive definite
      const double x0 = r0 * i0;
      const double x1 = (r1 - x0 * l10) * i1;
      const double x2 = (r2 - x0 * l20 - x1 * l21) * i2;
      const double x3 = (r3 - x0 * l30 - x1 * l31 - x2 * l32) * i3;
      P[lane * kPStride + 0] = x0; P[lane * kPStride + 1] = x1; P[lane * kPStride + 2] = x2; P[lane * kPStride + 3] = x3;
      if (lane < p) {                      // L_A column block, diagonal included (x_m at lane K+m is sqrt of the pivot)
        double* __restrict__ arow = A + lane * kLdA + K;
        if (lane >= K) arow[0] = x0;
        if (nb > 1 && lane >= K + 1) arow[1] = x1;
        if (nb > 2 && lane >= K + 2) arow[2] = x2;
        if (nb > 3 && lane >= K + 3) arow[3] = x3;
      }
      if (lane == 0) { rinv[K] = i0; if (nb > 1) rinv[K + 1] = i1; if (nb > 2) rinv[K + 2] = i2; if (nb > 3) rinv[K + 3] = i3; }
      __builtin_amdgcn_wave_barrier();
      switch (tjK) {
        case 0: reg_trailing<0>(c, P, nb, lane, (K + 4) >> 4); break;
        case 1: reg_trailing<1>(c, P, nb, lane, (K + 4) >> 4); break;
        case 2: reg_trailing<2>(c, P, nb, lane, (K + 4) >> 4); break;
        default: reg_trailing<3>(c, P, nb, lane, (K + 4) >> 4); break;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  if (slab == 0)
    for (int i = tid; i < p * p; i += 256) { const int r = i / p, c = i - r * p; Lp[i] = A[r * kLdA + c]; }
  // one row per thread: x L_A^T = b  <=>  x_j = (b_j - sum_{k<j} x_k L_A[j][k]) / L_A[j][j]
  const int r = p + slab * 256 + tid;
  if (r < fa) {
    double x[kMaxPiv];
    double* __restrict__ row = Lp + (size_t)r * p;
#pragma unroll
    for (int j = 0; j < kMaxPiv; j++) x[j] = j < p ? row[j] : 0.0;
#pragma unroll
    for (int j = 0; j < kMaxPiv; j++) {
      if (j < p) {
        double acc = x[j];
#pragma unroll
        for (int k = 0; k < j; k++) acc -= x[k] * A[j * kLdA + k];
        x[j] = acc * rinv[j];
      }
    }
#pragma unroll
    for (int j = 0; j < kMaxPiv; j++) if (j < p) row[j] = x[j];
  }
}

// ---- trailing update: U -= L_B L_B^T on 64x64 tiles of the lower triangle; wave w owns 16 rows, 4 MFMA tiles ---------
// The two 64 x p row panels of the tile are contiguous in L (row-major, stride p): staged into LDS with coalesced
// loads, MFMA operands gathered from there.
__global__ __launch_bounds__(256) void k_dense_trailing(DevGraph d, int level_begin, const int* __restrict__ off, int count) {
  extern __shared__ double pan[];
  int item;
  const int s = d.level_fronts[level_begin + locate(off, count, blockIdx.x, &item)];
  const int p = d.f_p[s], b = d.f_b[s], b1 = b + 1;
  const int ti = tri_row(item), tj = item - ti * (ti + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l16 = lane & 15, lq = lane >> 4;
  const int ld = p + 1;
  const double* __restrict__ Lb = d.L + d.f_Loff[s] + (size_t)p * p;     // rows p.. of the panel = L_B (+ rhs row)
  double* __restrict__ PA = pan;
  double* __restrict__ PB = (ti == tj) ? pan : pan + 64 * ld;
  {
    const int ra = ti * 64, na = min(64, b1 - ra) * p;
    const double* __restrict__ src = Lb + (size_t)ra * p;
    for (int i = tid; i < 64 * p; i += 256) { const int r = i / p, k = i - r * p; PA[r * ld + k] = i < na ? src[i] : 0.0; }
    if (ti != tj) {
      const int rb = tj * 64, nb = min(64, b1 - rb) * p;
      const double* __restrict__ src2 = Lb + (size_t)rb * p;
      for (int i = tid; i < 64 * p; i += 256) { const int r = i / p, k = i - r * p; PB[r * ld + k] = i < nb ? src2[i] : 0.0; }
    }
  }
  __syncthreads();
  const int R0 = ti * 64 + 16 * w, C0 = tj * 64;
  double4_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; t++) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
  if (R0 < b1) {
    const double* __restrict__ pa = PA + (16 * w + l16) * ld + lq;
    const double* __restrict__ pb = PB + l16 * ld + lq;
    for (int k0 = 0; k0 < p; k0 += 4) {
      const bool kok = k0 + lq < p;
      const double araw = pa[kok ? k0 : 0];
      const double av = kok ? -araw : 0.0;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const double braw = pb[16 * t * ld + (kok ? k0 : 0)];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, kok ? braw : 0.0, acc[t], 0, 0, 0);
      }
    }
  }
  // children's update matrices, pulled through inverse maps of this tile's 64 rows / 64 columns (front-local
  // indices p + ...); U is then written exactly once
  __shared__ int invr[64], invc[64];
  for (int ci = d.f_child_off[s]; ci < d.f_child_off[s + 1]; ci++) {
    const int c = d.child[ci];
    const int bc1 = d.f_b[c] + 1;
    const int* __restrict__ cm = d.cmap + d.f_cmap_off[c];
    const double* __restrict__ Uc = d.U + d.f_Uoff[c];
    __syncthreads();
    if (tid < 64) { invr[tid] = -1; invc[tid] = -1; }
    __syncthreads();
    const int fr0 = p + ti * 64, fc0 = p + tj * 64;
    for (int i = tid; i < bc1; i += 256) {
      const int m = cm[i];
      if (m >= fr0 && m < fr0 + 64) invr[m - fr0] = i;
      if (m >= fc0 && m < fc0 + 64) invc[m - fc0] = i;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int ic = invc[16 * t + l16];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int ir = invr[16 * w + lq + 4 * r];
        if (ir >= 0 && ic >= 0 && C0 + 16 * t + l16 <= R0 + lq + 4 * r) {
          const int hi = ir > ic ? ir : ic, lo = ir > ic ? ic : ir;
          acc[t][r] += Uc[(size_t)hi * bc1 + lo];
        }
      }
    }
  }
  double* __restrict__ Us = d.U + d.f_Uoff[s];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int cc = C0 + 16 * t + l16;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int rr = R0 + lq + 4 * r;                       // D layout: row (lane/16) + 4r, column lane%16
      if (rr < b1 && cc <= rr) Us[(size_t)rr * b1 + cc] = acc[t][r];
    }
  }
}

// ---- back-substitution of one level: x_p = L_A^-T (y - L_B^T x_b) --------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense_solve(DevGraph d, int level_begin) {
  __shared__ double A[kMaxPiv * kLdA];
  __shared__ double part[4][kMaxPiv];
  const int s = d.level_fronts[level_begin + blockIdx.x];
  const int p = d.f_p[s], b = d.f_b[s], f = p + b;
  const double* __restrict__ Lp = d.L + d.f_Loff[s];
  const int* __restrict__ bi = d.bidx + d.f_bidx_off[s];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < p * p; i += 256) { const int r = i / p, c = i - r * p; A[r * kLdA + c] = Lp[i]; }
  // t_k = y_k - sum_i L_B[i][k] x_b[i]: x_b staged once (one gather round trip), then every wave streams its
  // share of the rows (row i = p contiguous doubles, lane = column k), 8 independent loads in flight
  extern __shared__ double xb[];
  for (int i = tid; i < b; i += 256) xb[i] = d.delta[bi[i]];
  __syncthreads();
  double acc = 0.0;
  if (lane < p) {
    const double* __restrict__ Lb = Lp + (size_t)p * p + lane;
    int i = w;
    for (; i + 28 < b; i += 32) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = Lb[(size_t)(i + 4 * u) * p];
#pragma unroll
      for (int u = 0; u < 8; u++) acc -= v[u] * xb[i + 4 * u];
    }
    for (; i < b; i += 4) acc -= Lb[(size_t)i * p] * xb[i];
    if (w == 0) acc += Lp[(size_t)f * p + lane];
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w != 0) return;
  double t = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
  // single wave: lane k owns t_k; x_k is broadcast with v_readlane, no barrier
  for (int k = p - 1; k >= 0; k--) {
    const double cand = t / A[k * kLdA + k];
    const int lo = __builtin_amdgcn_readlane(__double2loint(cand), k), hi = __builtin_amdgcn_readlane(__double2hiint(cand), k);
    const double xk = __hiloint2double(hi, lo);
    if (lane == k) t = xk;
    else if (lane < k) t -= A[k * kLdA + lane] * xk;
  }
  if (lane < p) d.delta[d.pidx[d.f_poff[s] + lane]] = t;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

int dense_front_max_pivots() { return kMaxPiv; }

hipError_t launch_dense_hpush(const DevGraph& d, int max_el_per_front, double lambda, hipStream_t st) {
  if (d.n_fronts == 0) return hipSuccess;
  const int gx = std::max(1, std::min(64, cdiv(max_el_per_front, 256)));
  PPS_LAUNCH(k_dense_hpush, dim3(gx, d.n_fronts), dim3(256), 0, st, d, lambda);
  return hipGetLastError();
}

hipError_t launch_dense_factor_level(const DevGraph& d, int level_begin, int level_count, const int* off_asm, int n_asm,
                                     const int* off_pan, int n_pan, const int* off_trl, int n_trl, hipStream_t st) {
  if (level_count == 0) return hipSuccess;
  if (n_asm) PPS_LAUNCH(k_dense_assemble, dim3(n_asm), dim3(256), 0, st, d, level_begin, off_asm, level_count);
  if (n_pan) PPS_LAUNCH(k_dense_panel, dim3(n_pan), dim3(256), 0, st, d, level_begin, off_pan, level_count);
  if (n_trl) PPS_LAUNCH(k_dense_trailing, dim3(n_trl), dim3(256), (size_t)2 * 64 * (kMaxPiv + 1) * sizeof(double), st, d,
                                level_begin, off_trl, level_count);
  return hipGetLastError();
}

hipError_t launch_dense_solve_level(const DevGraph& d, int level_begin, int level_count, int level_max_b, hipStream_t st) {
  if (level_count == 0) return hipSuccess;
  PPS_LAUNCH(k_dense_solve, dim3(level_count), dim3(256), (size_t)std::max(1, level_max_b) * sizeof(double), st, d, level_begin);
  return hipGetLastError();
}

}  // namespace pps
